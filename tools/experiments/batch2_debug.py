"""Where does a B = 2 forward (un-padded fast path) leave the B = 1 forward of its first clip?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
from neurips2023_soc_amd.nested_tensor import NestedTensor  # noqa: E402

T, H, Wd, L = 8, 360, 640, 10
unpadded = os.environ.get("UNPADDED", "1") == "1"
same_text = os.environ.get("SAME_TEXT", "0") == "1"
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clips = [W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(2)]
ids = [W.synthetic_token_ids(1 + (0 if same_text else i), L).cuda() for i in range(2)]
st = {}
for B in (1, 2):
    clip = torch.stack(clips[:B], 1).contiguous()
    pad = torch.zeros(T, B, H, Wd, dtype=torch.bool, device="cuda")
    tok = torch.cat(ids[:B], 0)
    text = {"input_ids": tok, "attention_mask": torch.ones_like(tok)}
    targets = [[{"size": (H, Wd)}] * B for _ in range(T)]
    with torch.no_grad():
        sa = model.forward_backbone(NestedTensor(clip.clone(), pad, unpadded=unpadded), None, text)
        sb = model.forward_fuse_encode(sa)
        out = model.forward_tail(sb, targets)
    torch.cuda.synchronize()
    st[B] = (sa, sb, out)


def d(a, b):
    return float((a.float() - b.float()).abs().max())


(sa1, sb1, o1), (sa2, sb2, o2) = st[1], st[2]
for l in range(4):
    f1, f2 = sa1["feats"][l], sa2["feats"][l]          # '(b t) c h w'
    print("feats", l, tuple(f1.shape), tuple(f2.shape), d(f2.view(2, T, *f2.shape[1:])[0], f1.view(1, T, *f1.shape[1:])[0]))
print("words", d(sa2["words"][:, :1], sa1["words"]), "sentence", d(sa2["sentence"][:1], sa1["sentence"]))
m1, m2 = sb1["ctx"][0], sb2["ctx"][0]                   # memory '(b t) S c'
print("memory", tuple(m1.shape), tuple(m2.shape), d(m2.view(2, T, *m2.shape[1:])[0], m1.view(1, T, *m1.shape[1:])[0]))
print("lang_last", tuple(sb1["lang_last"].shape), tuple(sb2["lang_last"].shape), d(sb2["lang_last"][:, :1] if sb2["lang_last"].shape[1] == 2 else sb2["lang_last"][:1], sb1["lang_last"]))
for k in ("pred_masks", "pred_cls", "pred_boxes", "pred_logit"):
    a, b = o1[k], o2[k]
    print(k, tuple(a.shape), tuple(b.shape), d(b[:, :1] if b.dim() > 2 and b.shape[1] == 2 else b[:1], a))
