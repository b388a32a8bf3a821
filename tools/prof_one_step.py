import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
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
