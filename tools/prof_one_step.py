"""torch-profiler table of ONE backbone step (default FourCastNet C4; `python tools/prof_one_step.py C5_pangu_128x256x13`) grouped by input shape: finds the layout copies, clones and
elementwise passes torch inserts silently around custom kernels (how the 3 DtoD clones and 2 contiguous() copies per
block of the first version were found)."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tools"))
import bench_models as bm
from dlwp_benchmark_amd.weights import fill_state_dict
name = sys.argv[1] if len(sys.argv) > 1 else "C4_fourcastnet_128x256"
cls, cfg, batch, steps, (h, w) = bm.CONFIGS[name]
m = cls(**cfg); fill_state_dict(m, gain=0.7); m = m.to("cuda:0").eval()
cin = cfg["constant_channels"] + (cfg["prescribed_channels"] + cfg["prognostic_channels"]) * cfg.get("context_size", 1)
x = torch.randn(batch, cin, h, w, device="cuda:0")
with torch.no_grad():
    for _ in range(2): m.one_step(x)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
        m.one_step(x); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=40, max_shapes_column_width=70))
