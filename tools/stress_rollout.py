"""Repeats the headline rollout many times and checks every trajectory is bit-identical to the first one (the fused
kernel's hand-off protocol is timing dependent, its arithmetic is not): python tools/stress_rollout.py [repeats]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from dlwp_benchmark_amd.synthetic import navier_stokes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model, _ = build_model("cuda:0")
bad = 0
for batch in (32, 40, 5):
    prog = navier_stokes(batch, 21, 64, 64, seed=batch)[2].to("cuda:0")
    ref = model(prognostic=prog).clone()
    assert bool(torch.isfinite(ref).all())
    t0 = time.time()
    for i in range(n):
        out = model(prognostic=prog)
        if not torch.equal(out, ref):
            bad += 1
            print(f"batch {batch} repeat {i}: differs, max abs {float((out - ref).abs().max()):.3e}, "
                  f"finite {bool(torch.isfinite(out).all())}", flush=True)
    torch.cuda.synchronize()
    print(f"batch {batch}: {n} rollouts in {time.time() - t0:.1f} s, mismatches so far {bad}", flush=True)
sys.exit(1 if bad else 0)
