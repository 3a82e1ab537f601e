import torch, time
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
N=616*1024*1024  # floats -> 2.46 GB
x=torch.empty(N, device='cuda'); y=torch.empty(N, device='cuda'); z=torch.empty(N//6, device='cuda')
ms=t(lambda: x.fill_(1.0)); print("fill 2.46GB write-only: %.1f us  %.2f TB/s"%(ms*1e3, N*4/ms/1e9))
ms=t(lambda: y.copy_(x)); print("copy R+W: %.1f us  %.2f TB/s"%(ms*1e3, 2*N*4/ms/1e9))
ms=t(lambda: torch.sum(x)); print("sum read-only: %.1f us %.2f TB/s"%(ms*1e3, N*4/ms/1e9))
ms=t(lambda: torch.mul(x, 2.0, out=y)); print("mul R+W: %.1f us  %.2f TB/s"%(ms*1e3, 2*N*4/ms/1e9))
