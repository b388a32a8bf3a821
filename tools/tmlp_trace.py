#!/usr/bin/env python3
"""Reads a DLWP_TMLP_TRACE dump (token_mlp.hip): per wave of workgroup 0, s_memtime stamps at pass start, operands
ready, after every hidden-tile pair, plus s_memrealtime (100 MHz) at kernel start / end -> shader clock and the
per-phase cycle counts.   python tools/tmlp_trace.py gpurun_out/tmlp_trace.txt"""
import sys

rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
for r in rows:
    launch, wave, st = r[0], r[1], r[2:]
    if launch != 1:
        continue
    rt_entry, rt0, rt1 = st[253], st[254], st[255]
    s = [x for x in st[:253] if x]
    ncyc = s[-1] - s[0]
    clk = ncyc / ((rt1 - rt0) * 10e-9) / 1e9 if rt1 > rt0 else float("nan")
    per_pass = 2 + 8   # start, operands ready, 8 units
    npass = (len(s) - 1) // per_pass
    load, units = [], []
    for p in range(npass):
        b = s[p * per_pass:(p + 1) * per_pass + 1]
        load.append(b[1] - b[0])
        units.append([b[i + 1] - b[i] for i in range(1, 9)])
    flat = [u for us in units for u in us]
    print(f"wave {wave}: staging {(rt0 - rt_entry) * 0.01:.1f} us, loop {(rt1 - rt0) * 0.01:.1f} us; {npass} passes, {ncyc} cycles, clock {clk:.2f} GHz; operand phase avg {sum(load) / len(load):.0f} "
          f"(max {max(load)}); unit avg {sum(flat) / len(flat):.0f} min {min(flat)} max {max(flat)}; "
          f"last unit of a pass (incl. stores + next loop top) avg {sum(u[-1] for u in units) / len(units):.0f}")
    if wave in (0, 4):
        print("   pass 3 units:", units[3] if len(units) > 3 else units[-1])
