import torch, time
dev="cuda:0"
def timed(fn, reps=30):
    for _ in range(5): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
for (m,k,n) in [(65536,384,1152),(65536,384,1536),(65536,1536,384),(65536,384,384),(262144,192,576),(262144,192,768),(262144,768,192),(65536,96,288)]:
    x=torch.randn(m,k,device=dev,dtype=torch.bfloat16); w=torch.randn(n,k,device=dev,dtype=torch.bfloat16); b=torch.randn(n,device=dev,dtype=torch.bfloat16)
    t=timed(lambda: torch.nn.functional.linear(x,w,b))
    t2=timed(lambda: torch.matmul(x,w.t()))
    print(f"{m}x{k}->{n}: linear(bias) {t:.1f} us  matmul {t2:.1f} us  {2*m*k*n/t2/1e6:.0f} TF/s  bytes {(m*k+m*n)*2/t2/1e6:.2f} TB/s")
