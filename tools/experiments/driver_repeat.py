#!/usr/bin/env python
"""Run-to-run determinism of the Ref-YouTube-VOS driver: the same synthetic dataset through infer_refytb.run N times
(graphs on / off); every PNG must be byte-identical to the first run's.  Catches races in the decode / upload / replay /
writer pipeline that a single comparison with the oracle only sees when they happen to hit."""
import hashlib
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import infer_refytb, synthetic_dataset as SD, weights as W  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
tmp = tempfile.mkdtemp()
root = SD.make_dataset(os.path.join(tmp, "data"), videos=2, frames=3, height=144, width=256, expressions=2, seed=3)
tok = SD.HashTokenizer()


def digest(out_dir):
    h = {}
    for dp, _, files in os.walk(out_dir):
        for f in sorted(files):
            p = os.path.join(dp, f)
            h[os.path.relpath(p, out_dir)] = hashlib.md5(open(p, "rb").read()).hexdigest()
    return h


for graphs in (True, False):
    first, bad = None, 0
    for i in range(N):
        out = os.path.join(tmp, f"out_{int(graphs)}_{i}")
        infer_refytb.run(model, tok, root, out, size=96, max_size=160, decode_workers=2, use_graphs=graphs)
        d = digest(out)
        if first is None:
            first = d
        elif d != first:
            bad += 1
            diff = [k for k in d if d[k] != first.get(k)]
            print(f"graphs={graphs} run {i}: {len(diff)} of {len(d)} PNGs differ from run 0: {diff[:4]}", flush=True)
    print(f"graphs={graphs}: {bad} of {N - 1} repeat runs differ from the first ({len(first)} PNGs each)", flush=True)
