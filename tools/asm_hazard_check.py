#!/usr/bin/env python3
"""Build-time guard for the hazards hipcc does NOT handle around inline asm (run by __graft_entry__.build(); no GPU needed).

hipcc schedules an `asm volatile` statement as one opaque instruction: its hazard recogniser neither pads the wait
states of instructions inside the string nor looks at them when it pads its own code
(/opt/skills/guides/cdna_hip_programming.md section 5.7).  Round 1 hit that twice:

  A  `v_readlane_b32` / `v_readfirstlane_b32` (any VALU) writes an SGPR, an asm `global_load ... sc1` takes its base
     from that SGPR with fewer than 5 wait states in between -> load from a garbage address (GPU memory access
     fault; fixed in a196813 by `s_nop 4` opening the asm strings);
  B  a `v_mfma` result is read by an asm VALU instruction without the wait states an MFMA -> VALU dependency needs
     -> stale values, silently wrong columns (fixed in c4ee55f by making the first reads plain C);

and the guide names a third of the same class:

  C  an asm `*_store_dwordx3/x4` whose data registers the NEXT instruction overwrites (the store reads them late):
     needs `s_nop 1` closing the string.

Input: the device assembly the Makefile keeps beside every object (`-save-temps=obj`:
dlwp_benchmark_amd/csrc/build/*-hip-amdgcn-amd-amdhsa-gfx950.s), where `;;#ASMSTART` / `;;#ASMEND` delimit what came
from inline asm.  Only dependencies with at least ONE end inside such a block are checked -- everything else is the
compiler's own schedule, which its hazard recogniser covers -- so the wait-state table below can be conservative
without false alarms.  Wait states between two instructions = instructions issued in between, an `s_nop N` counting
N + 1 (the convention of the ISA's "required software-inserted wait states" tables).

Exit status 1 and one line per finding if anything is flagged.
"""
import glob
import os
import re
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Set, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SGPR_TO_VMEM_STATES = 5      # VALU writes SGPR -> VMEM reads that SGPR (guide 5.7 item 2: `s_nop 4`)
STORE_DATA_STATES = 2        # asm store x3/x4 -> overwrite of its data VGPRs (guide 5.7 item 1: `s_nop 1`)


def mfma_states(mnemonic: str) -> int:
    """wait states an MFMA's destination needs before anything but an accumulate-chain MFMA touches it: passes + 4
    (guide 5.7 item 2: "8-pass XDL: 12 states"); passes by output tile, taken on the conservative side."""
    m = re.match(r"v_mfma_\w+?_(\d+)x(\d+)x(\d+)", mnemonic)
    if not m:
        return 20
    rows = int(m.group(1))
    passes = {4: 2, 16: 8, 32: 16}.get(rows, 16)
    return passes + 4


REG_RE = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]")
VMEM_PREFIX = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_")


def regs_of(text: str) -> Set[Tuple[str, int]]:
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    if re.search(r"\bvcc\b", text):
        out.add(("s", 106)), out.add(("s", 107))
    return out


@dataclass
class Ins:
    line: int
    text: str
    mnemonic: str
    ops: List[str]
    in_asm: bool
    states: int = 1
    dst: Set[Tuple[str, int]] = field(default_factory=set)
    src: Set[Tuple[str, int]] = field(default_factory=set)


def split_ops(rest: str) -> List[str]:
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def n_dst_operands(mn: str) -> int:
    if mn.startswith(("s_nop", "s_waitcnt", "s_barrier", "s_sleep", "s_endpgm", "s_branch", "s_cbranch", "s_setprio",
                      "s_sethalt", "s_trap", "buffer_wbl2", "buffer_inv", "s_dcache", "s_icache")):
        return 0
    if mn.startswith(VMEM_PREFIX) or mn.startswith("ds_"):
        if "store" in mn or "write" in mn or ("atomic" in mn and "_rtn" not in mn) or " lds" in mn:
            return 0
        return 1
    if mn.startswith(("v_cmpx", "s_cmp", "s_bitcmp", "s_setpc", "s_cmpk")):
        return 0
    if mn.startswith(("v_readlane", "v_readfirstlane")):
        return 1
    if mn.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64_u32",
                      "v_mad_i64_i32", "v_div_scale")):
        return 2
    return 1


def parse_function(lines: List[Tuple[int, str]]) -> List[Ins]:
    out, in_asm = [], False
    for ln, raw in lines:
        t = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if raw.strip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.strip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.endswith(":") or t.startswith("."):
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        ops = split_ops(rest)
        ins = Ins(ln, t, mn, ops, in_asm)
        if mn == "s_nop":
            try:
                ins.states = int(ops[0], 0) + 1
            except Exception:
                ins.states = 1
        nd = n_dst_operands(mn)
        for i, op in enumerate(ops):
            (ins.dst if i < nd else ins.src).update(regs_of(op))
        if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
            # D = A * B + C: operand 0 is the destination only (C is operand 3)
            pass
        if mn.startswith("v_pk_fma") or mn.startswith("v_fma") or mn.startswith("v_mac") or mn.startswith("v_fmac"):
            if mn.startswith(("v_mac", "v_fmac")) and ops:
                ins.src.update(regs_of(ops[0]))
        out.append(ins)
    return out


def functions(path: str):
    """yields (name, [(line number, text)]) per function of a device .s file"""
    name, buf = None, []
    with open(path) as f:
        for i, raw in enumerate(f, 1):
            s = raw.rstrip("\n")
            m = re.match(r"^([A-Za-z_$][\w$.]*):\s*(;.*)?$", s)
            if m and not s.startswith(".L") and not s.startswith("\t"):
                if name and buf:
                    yield name, buf
                name, buf = m.group(1), []
                continue
            if s.strip().startswith(".end_amdhsa_kernel") or s.strip().startswith(".Lfunc_end"):
                if name and buf:
                    yield name, buf
                name, buf = None, []
                continue
            if name is not None:
                buf.append((i, s))
    if name and buf:
        yield name, buf


def is_valu(mn: str) -> bool:
    return mn.startswith("v_") and not mn.startswith(("v_mfma", "v_smfmac", "v_accvgpr"))


def is_vmem(mn: str) -> bool:
    return mn.startswith(VMEM_PREFIX)


def check_instructions(ins: List[Ins], where: str) -> List[str]:
    findings = []
    n = len(ins)
    for i, a in enumerate(ins):
        # ---- rule A: VALU writes SGPR -> VMEM reads it
        sdst = {r for r in a.dst if r[0] == "s"}
        if is_valu(a.mnemonic) and sdst:
            states = 0
            for j in range(i + 1, n):
                b = ins[j]
                if states >= SGPR_TO_VMEM_STATES:
                    break
                if is_vmem(b.mnemonic) and (b.src & sdst) and (a.in_asm or b.in_asm):
                    findings.append(f"{where}:{b.line}: [A] `{b.text}` reads an SGPR written by `{a.text}` (line {a.line}) "
                                    f"after {states} wait state(s); {SGPR_TO_VMEM_STATES} needed -> open the asm string with "
                                    f"`s_nop {SGPR_TO_VMEM_STATES - 1 - states}`")
                if b.dst & sdst:
                    break       # rewritten: later readers depend on that writer
                states += b.states
        # ---- rule B: MFMA destination -> anything but the accumulate chain
        if a.mnemonic.startswith(("v_mfma", "v_smfmac")) and a.ops:
            d = regs_of(a.ops[0])
            need = mfma_states(a.mnemonic)
            states = 0
            for j in range(i + 1, n):
                b = ins[j]
                if states >= need:
                    break
                touches = (b.src | b.dst) & d
                if touches:
                    chain = b.mnemonic.startswith(("v_mfma", "v_smfmac")) and len(b.ops) >= 4 and regs_of(b.ops[3]) == d \
                        and not (regs_of(b.ops[1]) | regs_of(b.ops[2])) & d
                    if not chain and (a.in_asm or b.in_asm):
                        findings.append(f"{where}:{b.line}: [B] `{b.text}` touches the destination of `{a.text}` (line {a.line}) "
                                        f"after {states} wait state(s); {need} needed -> pass the value through a compiler-visible "
                                        f"op first, or pad with s_nop inside the asm string")
                    if b.dst & d and not chain:
                        break
                    if chain:
                        break   # the chain's own result is the next producer (checked from its own index)
                states += b.states
        # ---- rule C: asm wide store -> overwrite of its data registers
        if a.in_asm and is_vmem(a.mnemonic) and re.search(r"store_dwordx[34]", a.mnemonic):
            data = set()
            for op in a.ops:
                r = regs_of(op)
                if len(r) >= 3 and all(k == "v" for k, _ in r):
                    data = r
            states = 0
            for j in range(i + 1, n):
                b = ins[j]
                if states >= STORE_DATA_STATES:
                    break
                if (b.dst & data) and not b.mnemonic.startswith("s_"):
                    findings.append(f"{where}:{b.line}: [C] `{b.text}` overwrites data registers of the asm store `{a.text}` "
                                    f"(line {a.line}) after {states} wait state(s); {STORE_DATA_STATES} needed -> close the asm "
                                    f"string with `s_nop {STORE_DATA_STATES - 1 - states}`")
                    break
                states += b.states
    return findings


def check_file(path: str) -> Tuple[List[str], int, int]:
    findings, n_fn, n_asm = [], 0, 0
    for name, lines in functions(path):
        ins = parse_function(lines)
        if not ins:
            continue
        n_fn += 1
        n_asm += sum(1 for x in ins if x.in_asm)
        findings += check_instructions(ins, f"{os.path.basename(path)}({name[:60]})")
    return findings, n_fn, n_asm


def check_text(text: str, where: str = "<text>") -> List[str]:
    """checks one instruction stream given as text (unit tests: the pre-fix patterns must be flagged)"""
    lines = [(i, s) for i, s in enumerate(text.splitlines(), 1)]
    return check_instructions(parse_function(lines), where)


def main(argv: Optional[List[str]] = None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    paths = argv or sorted(glob.glob(os.path.join(ROOT, "dlwp_benchmark_amd", "csrc", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not paths:
        print("asm_hazard_check: no device assembly found (build with the Makefile: -save-temps=obj keeps it)", file=sys.stderr)
        return 2
    total, n_fn, n_asm = [], 0, 0
    for p in paths:
        f, a, b = check_file(p)
        total += f
        n_fn += a
        n_asm += b
    for line in total:
        print(line)
    print(f"asm_hazard_check: {len(paths)} files, {n_fn} functions, {n_asm} inline-asm instructions, {len(total)} finding(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
