import sys, torch
sys.path.insert(0,'.')
from neurips2023_soc_amd import hot_ops
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)*1e3/reps
gen=torch.Generator().manual_seed(0)
for (M,N,K,cuts) in ((38560,256,256,[None,(128,2),(256,1)]),(32768,256,256,[None,(256,1),(128,2)]),(5792,256,256,[None,(64,4),(46,4),(91,2),(32,8)]),
                     (38560,384,256,[None]),(32768,384,256,[None,(256,1)]),(5792,384,256,[None,(64,4)])):
    x=torch.randn(M,K,generator=gen).cuda(); w=(torch.randn(N,K,generator=gen)/16).cuda(); b=torch.randn(N,generator=gen).cuda()
    for cut in cuts:
        try:
            us=t(lambda: hot_ops.xs_linear(x,w,b,None,None,"none",cut=cut))
            print(M,N,K,"cut",cut,"plan",hot_ops.xs_linear_plan(M,N,K) if cut is None else "", "%.1f us"%us, flush=True)
        except Exception as ex:
            print(M,N,K,cut,"ERR",str(ex)[:80])
    us=t(lambda: hot_ops.ws_linear(x,w,b) if hasattr(hot_ops,'ws_linear') else None)
    print(M,N,K,"K13b %.1f us"%us)
