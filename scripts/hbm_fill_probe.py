import torch
dev=torch.device('cuda:0')
x=torch.empty(944*1024*1024//2, dtype=torch.bfloat16, device=dev)
y=torch.empty_like(x)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
ms=t(lambda: x.fill_(1.0)); print('fill 944 MiB: %.1f us  %.2f TB/s'%(ms*1e3, x.numel()*2/ms/1e9))
ms=t(lambda: x.zero_()); print('zero 944 MiB: %.1f us  %.2f TB/s'%(ms*1e3, x.numel()*2/ms/1e9))
ms=t(lambda: y.copy_(x)); print('copy 944 MiB: %.1f us  %.2f TB/s (read+write)'%(ms*1e3, 2*x.numel()*2/ms/1e9))
ms=t(lambda: torch.add(x,1,out=y)); print('add  944 MiB: %.1f us  %.2f TB/s (read+write)'%(ms*1e3, 2*x.numel()*2/ms/1e9))
