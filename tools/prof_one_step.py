"""torch-profiler table of ONE FourCastNet (C4) step grouped by input shape: finds the layout copies, clones and
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
cls, cfg, batch, steps, (h, w) = bm.CONFIGS["C4_fourcastnet_128x256"]
m = cls(**cfg); fill_state_dict(m, gain=0.7); m = m.to("cuda:0").eval()
x = torch.randn(batch, 8, h, w, device="cuda:0")
with torch.no_grad():
    for _ in range(2): m.one_step(x)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
        m.one_step(x); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=40, max_shapes_column_width=70))
