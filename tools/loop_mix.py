"""Instruction mix of the loops of one function in a device assembly file (-save-temps .s).
usage: python tools/loop_mix.py <file.s> <mangled-name-substring> [must-contain-op]"""
import collections
import re
import sys

path, fname = sys.argv[1], sys.argv[2]
need = sys.argv[3] if len(sys.argv) > 3 else None
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(fname + ':'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
fn = lines[start:end]
labels = {}
for i, l in enumerate(fn):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(fn):
    m = re.search(r'\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))


def cat(op):
    if op.startswith('v_mfma') or op.startswith('v_pk_'):
        return op
    if op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_log', 'v_sqrt')):
        return 'v_trans'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'ds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return 'vmem'
    if op.startswith('s_nop'):
        return 's_nop'
    if op.startswith('s_waitcnt'):
        return 's_waitcnt'
    return 'salu'


for a, b in loops:
    body = fn[a:b + 1]
    ops = [x.split()[0] for x in body if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    if need and not any(o.startswith(need) for o in ops):
        continue
    c = collections.Counter(cat(o) for o in ops)
    print(f"loop lines {a + start + 1}-{b + start + 1}: {len(ops)} instrs", dict(c))
    vo = collections.Counter(o for o in ops if cat(o) == 'valu')
    print('    valu:', vo.most_common(16))
    print('    nops:', collections.Counter(x.strip() for x in body if 's_nop' in x))
