"""K25 (small_attn.hip) against torch's scaled_dot_product_attention (AOTriton attn_fwd), 50 calls per graph replay."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neurips2023_soc_amd import hot_ops
import torch.nn.functional as F
for (B,L,H,D) in ((1,10,12,64),(1,24,12,64),(2,32,12,64),(1,64,12,64)):
    q,k,v=(torch.randn(B,L,H*D,device="cuda") for _ in range(3))
    def a(): return hot_ops.small_attention(q,k,v,H)
    q4,k4,v4=(t.view(B,L,H,D).transpose(1,2) for t in (q,k,v))
    def b(): return F.scaled_dot_product_attention(q4,k4,v4)
    for name,fn in (("k25",a),("sdpa",b)):
        g=torch.cuda.CUDAGraph()
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(50): fn()
        g.replay(); torch.cuda.synchronize()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        print(B,L,name, "%.2f us per call in a graph of 50"%(s.elapsed_time(e)*1e3/50))
