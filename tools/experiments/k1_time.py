import sys, torch
sys.path.insert(0,'.')
from neurips2023_soc_amd import hot_ops
def timeit(fn,n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for name,(D,H,W,nH) in {"s0":(8,90,160,3),"s1":(8,45,80,6),"s2":(8,23,40,12),"s3":(8,12,20,24)}.items():
    C=32*nH; g=torch.Generator(device='cuda').manual_seed(1)
    qkv=torch.randn(1,D,H,W,3*C,device='cuda',generator=g); qb=torch.randn(3*C,device='cuda',generator=g); tab=torch.randn(2535,nH,device='cuda',generator=g)
    print(name, [round(timeit(lambda: hot_ops.window_attention3d(qkv,qb,tab,nH,(8,7,7),sh)),1) for sh in ((0,0,0),(4,3,3))])
