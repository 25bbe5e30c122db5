"""Record the best hipBLASLt/rocBLAS kernel for every GEMM shape of the SOC forward (run on an
MI355X):   python tools/tune_gemms.py [--backbone video-swin-t] [--out gpurun_out/tunableop_gfx950.csv]
The result is copied to neurips2023_soc_amd/tunableop_gfx950.csv and committed."""
import argparse
import os
import sys

os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "120")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "10")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/tunableop_gfx950.csv")
ap.add_argument("--configs", default="video-swin-t:8:360:640,video-swin-b:8:360:640")
a = ap.parse_args()
tunable.set_filename(a.out)

import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402

for cfg in a.configs.split(","):
    bb, T, H, Wd = cfg.split(":")
    T, H, Wd = int(T), int(H), int(Wd)
    model, _, _ = S.build_model(S.default_args(bb, text_encoder_random_init=True))
    W.load_synthetic(model, 2023)
    model = model.cuda().eval()
    clip = W.synthetic_clip(1, T, H, Wd).cuda()
    ids = W.synthetic_token_ids(1, 10).cuda()
    for _ in range(2):
        samples = S.NestedTensor(clip[:, None], torch.zeros(T, 1, H, Wd, dtype=torch.bool, device="cuda"))
        model(samples, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)}, [[{"size": (H, Wd)}]] * T)
    torch.cuda.synchronize()
    print("tuned", cfg, flush=True)
    del model
# TunableOp writes the file itself at interpreter exit (set_filename above)
print("wrote", a.out)
