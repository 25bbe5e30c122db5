"""Which part of the tail costs the pipelined replay its time?  Pipelined per-clip time with parts of the tail replaced by
cached constants (timing only: the results of the ablated variants are wrong by construction)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.graph_runner import PipelinedClipGraph  # noqa: E402

T, H, Wd, L = 8, 360, 640, 10
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clip = W.synthetic_clip(1, T, H, Wd).cuda()
ids = W.synthetic_token_ids(1, L).cuda()


def run(n=40):
    with torch.no_grad():
        g = PipelinedClipGraph(model, T, H, Wd, L, "cuda")
        for _ in range(4):
            g.run(clip, ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.run(clip, ids)
        g.flush()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


print(f"full tail                          {run():.3f} ms/clip", flush=True)

# (a) no FPN: a constant map instead of the spatial decoder
sd_forward = model.spatial_decoder.forward
fpn_const = torch.zeros(T, 8, 90, 160, device="cuda")
model.spatial_decoder.forward = lambda x, feats: fpn_const
print(f"without the FPN spatial decoder    {run():.3f} ms/clip", flush=True)
model.spatial_decoder.forward = sd_forward

# (b) no decoder value projections: cached projected values
tr = model.transformer
orig_decode = tr.decode
cache = {}


def decode_cached(ctx, tgt, query_embed, values=None):
    if "v" not in cache:
        memory = ctx[0]
        cache["v"] = [layer.cross_attn.value_proj(memory) for layer in tr.decoder.layers]
    return orig_decode(ctx, tgt, query_embed, values=cache["v"])


tr.decode = decode_cached
print(f"without decoder value projections  {run():.3f} ms/clip", flush=True)
model.spatial_decoder.forward = lambda x, feats: fpn_const
print(f"without both                       {run():.3f} ms/clip", flush=True)

# (c) no memory_maps copies either
orig_maps = tr.memory_maps
mm = {}
tr.memory_maps = lambda ctx: mm.setdefault("m", orig_maps(ctx))
print(f"... and without memory_maps copies {run():.3f} ms/clip", flush=True)

# (d) empty tail
model.forward_tail_orig = model.forward_tail
out_const = {}


def tail_const(state, targets, fork=True):
    if "o" not in out_const:
        out_const["o"] = model.forward_tail_orig(state, targets, fork=False)
    return out_const["o"]


model.forward_tail = tail_const
print(f"tail = selection + record only     {run():.3f} ms/clip", flush=True)
