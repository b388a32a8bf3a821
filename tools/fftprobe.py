import torch
x = torch.randn(4, 128, 256, 64, device="cuda")
xf = torch.fft.rfft2(x, dim=(1, 2), norm="ortho")
print("rfft2 out", xf.shape, xf.stride(), xf.is_contiguous())
y = torch.fft.irfft2(xf, s=(128, 256), dim=(1, 2), norm="ortho")
print("irfft2 out", y.shape, y.stride(), y.is_contiguous())
# channels-first alternative
xc = x.permute(0, 3, 1, 2).contiguous()
xfc = torch.fft.rfft2(xc, norm="ortho")
print("NCHW rfft2 out", xfc.shape, xfc.stride(), xfc.is_contiguous())
from torch.profiler import profile, ProfilerActivity
for name, fn in (("nhwc", lambda: torch.fft.irfft2(torch.fft.rfft2(x, dim=(1,2), norm="ortho"), s=(128,256), dim=(1,2), norm="ortho")),
                 ("nchw", lambda: torch.fft.irfft2(torch.fft.rfft2(xc, norm="ortho"), s=(128,256), norm="ortho"))):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    print(name)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=60))
