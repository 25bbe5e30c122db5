"""Do HIP stream priorities reach the branches of the pipelined graph?  (experiment, result in README.md: no)

The capture-priority rows need this two-line patch of graph_runner.PipelinedClipGraph.__init__ (not in the tree):
    self._pc = torch.cuda.Stream(device=dev, priority=self.TAIL_PRIORITY)
    cap = torch.cuda.Stream(device=dev, priority=self.HEAD_PRIORITY);  with torch.cuda.graph(g, stream=cap): ...
Without it only the last row (launching the replays from a high-priority stream) measures anything."""
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


def run(stream, n=40):
    with torch.cuda.stream(stream), torch.no_grad():
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


print("priority range", torch.cuda.Stream.priority_range())
for hp, tp in ((0, 0), (-1, 0), (0, -1), (-1, -1)):
    PipelinedClipGraph.HEAD_PRIORITY, PipelinedClipGraph.TAIL_PRIORITY = hp, tp
    print(f"capture priorities head {hp:2d} tail {tp:2d}: launched from the default stream {run(torch.cuda.current_stream()):.3f} ms/clip",
          flush=True)
PipelinedClipGraph.HEAD_PRIORITY = PipelinedClipGraph.TAIL_PRIORITY = 0
print(f"launched from a high-priority stream  {run(torch.cuda.Stream(priority=-1)):.3f} ms/clip")
