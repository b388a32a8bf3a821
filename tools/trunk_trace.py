"""Phase timeline of fno_trunk_kernel from a DLWP_TRUNK_TRACE dump (100 MHz s_memrealtime stamps).
usage: DLWP_TRUNK_TRACE=gpurun_out/trunk_trace.txt python tools/trunk_trace.py gpurun_out/trunk_trace.txt"""
import os
import sys

import numpy as np

path = sys.argv[1]
if not os.path.exists(path) or os.environ.get("DLWP_TRUNK_TRACE"):
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_model
    from dlwp_benchmark_amd.synthetic import navier_stokes

    model, _ = build_model("cuda:0")
    prog = navier_stokes(32, 4, 64, 64, seed=1)[2].to("cuda:0")
    for _ in range(3):   # the first launch of a process pays cold instruction caches; look at the later ones
        model(prognostic=prog)
    torch.cuda.synchronize()
rows = [list(map(int, l.split())) for l in open(path)]
names = ["start"]
if os.environ.get("DLWP_FNO_STEP", "1") != "0":
    names = ["step top", "lift staged", "lift seg", "lift all waves", "lift done", "trunk setup"]
for l in range(4):
    names += [f"L{l} y-ready", f"L{l} P1 done", f"L{l} barrier1", f"L{l} P2 done", f"L{l} barrier2", f"L{l} P3+sync",
              f"L{l} skip+idft"]
    if l < 3:
        names += [f"L{l} gelu", f"L{l} transpose", f"L{l} fwd dft"]
names += ["rows done", "proj done"]
for launch in sorted({r[0] for r in rows})[-int(os.environ.get('TRACE_LAUNCHES', '1')):]:
    t = np.array([r[2:] for r in rows if r[0] == launch], dtype=np.float64)
    t0 = t[:, 0].min()
    rel = (t - t0) / 100.0
    print(f"launch {launch}: {t.shape[0]} workgroups, span {rel.max():.2f} us; start skew max {rel[:, 0].max():.2f} us")
    prev = rel[:, 0]
    for k in range(1, t.shape[1]):
        d = rel[:, k] - rel[:, k - 1]
        print(f"  {names[k % len(names)]:>14}: at mean {rel[:, k].mean():6.2f} us | phase mean {d.mean():5.2f} min {d.min():5.2f} max {d.max():5.2f}")
