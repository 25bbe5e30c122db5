"""Files -> PNGs throughput of the Ref-YouTube-VOS driver (reference infer_refytb.py:193-207,269-277) and the host budget behind
it: the driver on a synthetic Ref-YouTube-VOS-shaped set (N videos x 8 frames of 720p JPEG x 3
expressions), plus the CPU seconds one clip costs on the host side -- JPEG decode (once per video, shared by its expressions)
and PNG encode (8 masks per clip) -- measured single-threaded, and the clips/s at which the granted CPUs saturate.

    python tools/files_to_png.py [--videos 64] [--out profiles/r05_files_to_png.json]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=64)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_files_to_png.json"))
ap.add_argument("--group", type=int, default=8, help="clips per launch group of the driver's graph pipeline (the driver's default with --graphs: 8)")
ap.add_argument("--ragged", action="store_true", help="1-5 expressions per video (as the real set: infer_refytb.py:185), mean 3, instead of 3 each")
a = ap.parse_args()

import numpy as np  # noqa: E402

from neurips2023_soc_amd import clip_io, infer_refytb, synthetic_dataset  # noqa: E402
from neurips2023_soc_amd.clip_parallel import granted_cpus  # noqa: E402

tmp = tempfile.mkdtemp(prefix="soc_f2p_")
root, out_dir = os.path.join(tmp, "data"), os.path.join(tmp, "out")
counts = [(1, 3, 5, 2, 4, 3, 2, 4)[v % 8] for v in range(a.videos)] if a.ragged else 3          # mean 3 either way
synthetic_dataset.make_dataset(root, videos=a.videos, frames=8, expressions=counts, n_words=8)
# ---- the driver itself, warm pass reported (child process: its own GPU context, the parent stays CPU-only)
cmd = [sys.executable, "-m", "neurips2023_soc_amd.infer", "--dataset", "refytb", "--root", root, "--out", out_dir, "--graphs",
       "--repeat", "2", "--group", str(a.group)]
t0 = time.perf_counter()
r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
wall = time.perf_counter() - t0
lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
if r.returncode != 0 or not lines:
    print(r.stderr[-3000:])
    raise SystemExit(1)
stats = json.loads(lines[-1])
# ---- host cost per clip, one thread: decode 8 x 720p JPEG (one video), encode 8 x 720p 1-bit PNG (one clip)
img_folder, data = infer_refytb.load_meta(root, "valid")
video = sorted(data)[0]
paths = clip_io.frame_paths(img_folder, video, data[video]["frames"])
dec = []
for _ in range(5):
    c0 = time.process_time()
    frames = [clip_io.decode_frame(p) for p in paths]
    dec.append(time.process_time() - c0)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:720, 0:1280]
masks = [((yy - 360) ** 2 + (xx - 640 - 20 * j) ** 2 < (150 + 10 * j) ** 2) for j in range(8)]      # blob-shaped, like an object mask
enc = []
for _ in range(5):
    c0 = time.process_time()
    for j, m in enumerate(masks):
        infer_refytb.save_binary_mask(m, os.path.join(tmp, f"m{j}.png"))
    enc.append(time.process_time() - c0)
dec_s, enc_s = sorted(dec)[len(dec) // 2], sorted(enc)[len(enc) // 2]
cpu_s_per_clip = dec_s / 3.0 + enc_s                 # a video's frames are decoded once for its 3 expressions
cpus = granted_cpus()
res = {
    "driver": {k: stats[k] for k in ("videos", "expressions", "frames", "seconds", "clips_per_s", "seconds_input", "seconds_model",
                                     "seconds_writer_tail", "group_replays", "part_filled_replays", "stale_slots",
                                     "remainder_singles") if k in stats},
    "expressions_per_video": "1-5, ragged (mean 3)" if a.ragged else "3",
    "driver_command": " ".join(cmd[1:]), "driver_wall_s_two_passes_plus_start": wall,
    "host_cpu_seconds": {"jpeg_decode_8x720p_one_video": dec_s, "png_encode_8x720p_one_clip": enc_s,
                         "per_clip_at_3_expressions_per_video": cpu_s_per_clip},
    "granted_cpus": cpus,
    "host_bound_clips_per_s": cpus / cpu_s_per_clip,
    "note": "host_bound_clips_per_s = granted CPUs / CPU seconds per clip (decode shared by the video's expressions + 8 PNGs): "
            "the rate at which JPEG decode + PNG encode alone saturate this box's CPU quota; eight ranks share the same quota",
}
os.makedirs(os.path.dirname(a.out), exist_ok=True)
with open(a.out, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
