import sys, time, torch
sys.path.insert(0,'.')
import neurips2023_soc_amd as S
from neurips2023_soc_amd import weights as W
from neurips2023_soc_amd.graph_runner import ClipGraph
model,_,_=S.build_model(S.default_args(text_encoder_random_init=True)); W.load_synthetic(model,2023); model=model.cuda().eval()
g0=ClipGraph(model,8,360,640,32,'cuda')   # warm everything
for T,w in ((8,2),(20,2),(36,2),(20,1),(20,0)):
    torch.cuda.synchronize(); t=time.perf_counter()
    g=ClipGraph(model,T,360,640,32,'cuda',warmup=w)
    torch.cuda.synchronize(); print(f"T={T} warmup={w}: capture {1e3*(time.perf_counter()-t):.0f} ms", flush=True)
    clip=torch.randn(T,3,360,640,device='cuda'); ids=torch.ones(1,32,dtype=torch.long,device='cuda')
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): g.run(clip, ids)
    torch.cuda.synchronize(); print(f"   replay {1e3*(time.perf_counter()-t)/5:.1f} ms/clip", flush=True)
    del g
