"""Debug aid: dump one backbone step of the C2 model (python tools/step_diff.py out.pt), run once per setting of
DLWP_FNO_STEP, then `python tools/step_diff.py a.pt b.pt` prints where the two differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    d = (a - b).abs()
    print("max abs diff", float(d.max()), "ref max", float(b.abs().max()))
    per_row = d.amax(dim=(1, 3))          # [B, H]
    for s in range(a.shape[0]):
        bad = (per_row[s] > 1e-5 * float(b.abs().max())).nonzero().flatten().tolist()
        print(f"sample {s}: {len(bad)} bad rows", bad[:40])
    per_col = d.amax(dim=(0, 1, 2))
    print("bad columns", (per_col > 1e-5 * float(b.abs().max())).nonzero().flatten().tolist()[:70])
else:
    from bench import build_model
    from dlwp_benchmark_amd.synthetic import navier_stokes

    model, _ = build_model("cuda:0")
    x = navier_stokes(4, 1, 64, 64, seed=3)[2][:, 0].to("cuda:0")
    y = model.one_step(x)
    torch.cuda.synchronize()
    torch.save(y.cpu(), sys.argv[1])
