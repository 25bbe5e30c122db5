"""K24 at K = 768 with a LayerNorm in front: do the builds with 18 and with 12 column tiles per range give the same bits, with
and without the accumulation-register pin of the row fragments (tools/experiments/libsoc_hip_nopin.so = -DSOC_K24_NO_PIN)?
Identity weights: the output IS the kernel's normalised row.   python tools/experiments/k24_dbg.py [nopin]
The second library is not kept in the tree: compile xs_linear_split.hip and xs_linear_split_wide.hip with -DSOC_K24_NO_PIN and link
them with the other objects of csrc/_obj into tools/experiments/libsoc_hip_nopin.so first."""
import os
import subprocess
import sys
import torch
sys.path.insert(0, ".")
from neurips2023_soc_amd import _lib  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] == "nopin":
    _lib.LIB_PATH = os.path.abspath("tools/experiments/libsoc_hip_nopin.so")
from neurips2023_soc_amd import hot_ops  # noqa: E402
g = torch.Generator().manual_seed(0)
M, N, K = 1920, 1152, 768
x = torch.randn(M, K, generator=g).cuda()
w = torch.zeros(N, K).cuda(); w[:K] = torch.eye(K).cuda()
lnp = ((torch.rand(K, generator=g) + 0.5).cuda(), torch.randn(K, generator=g).cuda() * 0.1, 1e-5)
tag = "nopin" if len(sys.argv) > 1 else "pin"
res = {}
for ln in (None, lnp):
    for cut in ((30, 4), (30, 6), (30, 12)):
        o = hot_ops.xs_linear(x, w, None, ln, None, "none", cut=cut)[:, :K]
        res[("ln" if ln else "plain", N // 16 // cut[1])] = o.cpu()
        if ln is None:
            print(tag, "no LayerNorm, span", N // 16 // cut[1], "tiles: output == x exactly:", bool(torch.equal(o, x)))
torch.save({k: v for k, v in res.items() if k[0] == "ln"}, f"/tmp/k24_dbg_{tag}.pt")
if tag == "pin":
    subprocess.run([sys.executable, __file__, "nopin"], check=True)
    other = torch.load("/tmp/k24_dbg_nopin.pt")
    ref64 = torch.nn.functional.layer_norm(x.double(), (K,), lnp[0].double(), lnp[1].double(), 1e-5).cpu()
    for k in sorted(other, key=str):
        a, b = res[k], other[k]
        print(k, "pin vs nopin: differing elements", int((a != b).sum()), " pin err vs f64 %.3e  nopin err %.3e" % (
            float((a.double() - ref64).abs().max()) if k[0] == "ln" else 0.0, float((b.double() - ref64).abs().max()) if k[0] == "ln" else 0.0))
    for kind in ("ln",):
        for lib, r in (("pin", res), ("nopin", other)):
            print(lib, kind, "span 18 vs 12:", int((r[(kind, 18)] != r[(kind, 12)]).sum()), " 18 vs 6:", int((r[(kind, 18)] != r[(kind, 6)]).sum()))
    d = (res[("ln", 18)] != other[("ln", 18)]).nonzero()
    d2 = (res[("ln", 12)] != other[("ln", 12)]).nonzero()
    print("pin-18 vs nopin-18 first diffs", d[:6].tolist(), "pin-12 vs nopin-12", d2[:6].tolist())
    for (r, c) in (d[:4].tolist() or d2[:4].tolist()):
        print("  element", r, c, "x", float(x[r, c]), "pin18 %.9g nopin18 %.9g pin12 %.9g f64 %.12g" % (
            float(res[("ln", 18)][r, c]), float(other[("ln", 18)][r, c]), float(res[("ln", 12)][r, c]), float(ref64[r, c])))
