"""Ref-YouTube-VOS inference driver: the per-process body of the reference's infer_refytb.py
(`sub_processor`, :112-285) on the MI355X pipeline.

    for each video (this rank's share)            reference :160, split :84-109
      for each expression of the video            :181
        frames -> resized, normalised clip        :191-207   here: decoded once per VIDEO, K9 on the GPU
        forward, pick the best query              :209-226   SOC.forward (K1-K5, K7, K8) + select_trajectory
        up-sample to the original size, > 0.5     :230-231   K6
        write <out>/<video>/<exp_id>/<frame>.png  :269-277   8-bit 'L' PNG, 0 / 255, from a writer pool

Dataset layout (reference :62-82): <root>/valid/JPEGImages/<video>/<frame>.jpg and
<root>/meta_expressions/valid/meta_expressions.json = {"videos": {video: {"frames": [...],
"expressions": {exp_id: {"exp": str}}}}}.  Visualisation (:233-266) is not part of the hot path.
"""
from __future__ import annotations

import json
import os
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, Optional, Sequence

import numpy as np
import torch

from . import clip_io
from .infer import ClipInferencer

Tokenize = Callable[[str], torch.Tensor]


def split_videos(videos: Sequence[str], rank: int, world: int):
    """Contiguous shares, the remainder to the last rank (reference infer_refytb.py:92-99)."""
    per = len(videos) // world
    return list(videos[rank * per:] if rank == world - 1 else videos[rank * per:(rank + 1) * per])


def save_binary_mask(mask: np.ndarray, path: str) -> None:
    """bool/0-1 [H,W] -> 8-bit 'L' PNG with 0 / 255 (reference :272-277).  Same mode and pixels as the reference writes.
    Written by the run-length encoder of libsoc_host.so (png_fast.py: ~0.2 ms of CPU per 720p mask where zlib needs 1.9 at its
    fastest level and 3.0 at Pillow's default -- at 8 masks per clip the PNG encoder was the larger half of the host's work,
    profiles/r04_files_to_png.json); SOC_PNG=pillow writes through Pillow instead (SOC_PNG_LEVEL, default Pillow's own)."""
    from . import png_fast
    if png_fast.use_pillow():
        from PIL import Image
        Image.fromarray(mask.astype(np.uint8) * 255, mode="L").save(path, **png_fast.pillow_save_kwargs())
    else:
        png_fast.save(path, mask, binarize=True)


def load_meta(root: str, split: str = "valid"):
    img_folder = os.path.join(root, split, "JPEGImages")
    with open(os.path.join(root, "meta_expressions", split, "meta_expressions.json")) as f:
        return img_folder, json.load(f)["videos"]


PREFETCH_VIDEOS = 4          # videos decoded ahead of the one on the GPU (a 720p clip of 8 frames is 22 MB of pinned staging)


@torch.no_grad()
def run(model, tokenize: Tokenize, root: str, out_dir: str, rank: int = 0, world: int = 1, device="cuda",
        split: str = "valid", size: int = 360, max_size: Optional[int] = 640, use_graphs: bool = False,
        decode_workers: int = 8, writer_workers: int = 16, videos: Optional[Sequence[str]] = None,
        engine: Optional[ClipInferencer] = None, pad_tokens_to: Optional[int] = 32, group: int = 1) -> Dict:
    """Process this rank's videos; returns counters + timings.  group > 1 (with use_graphs): consecutive clips of one
    geometry share every launch of the forward, each still getting its single-clip result (ClipInferencer(group=...)).  `tokenize(expression) -> int64 [1,L]`
    (RobertaTokenizerFast in production; no vocabulary files exist offline, so the caller supplies it)."""
    img_folder, data = load_meta(root, split)
    todo = split_videos(sorted(data.keys()) if videos is None else list(videos), rank, world)
    engine = engine or ClipInferencer(model, device, use_graphs=use_graphs, pad_tokens_to=pad_tokens_to, group=group)
    cache = clip_io.VideoClipCache(clip_io.FramePreprocessor(device, size, max_size), workers=decode_workers)
    stats = {"videos": 0, "expressions": 0, "frames": 0, "seconds_input": 0.0, "seconds_model": 0.0}
    pending = []
    t0 = time.perf_counter()

    def flush(job):
        """clip i's masks -> PNG jobs; called after clip i+1 has been enqueued, so the D2H copy, the numpy
        view and the job submission overlap the next forward instead of idling the GPU"""
        host, done, save_dir, names = job
        done.synchronize()
        masks = host.numpy().copy()            # the pinned buffer is reused two clips later
        for j, name in enumerate(names):
            pending.append(writers.submit(save_binary_mask, masks[j], os.path.join(save_dir, name + ".png")))

    pinned: Dict = {}                      # two host buffers per mask shape, reused (pinning is slow)

    def host_buffer(shape, slot):
        key = (tuple(shape), slot)
        if key not in pinned:
            pinned[key] = torch.empty(tuple(shape), dtype=torch.bool, pin_memory=True)
        return pinned[key]

    d2h = torch.cuda.Stream(device=device)    # mask download overlaps the next clip's forward

    streaming = engine.use_graphs        # graphs: software-pipelined replays, results arrive one replay late
    # downloads left in flight behind the newest results: with a group pipeline (engine.group clips per replay) the results
    # of a group arrive together, right behind the replay of the NEXT group -- waiting for them at once would hold the host
    # for that whole replay
    lag = engine.group if streaming else 1

    def enqueue_download(res):
        """masks of a finished clip -> pinned host buffer on the d2h stream; returns the job for flush()"""
        masks, (save_dir, names) = res["masks"], res["tag"]
        host = host_buffer(masks.shape, enqueue_download.n % (2 * lag + 2))
        enqueue_download.n += 1
        d2h.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(d2h):
            host.copy_(masks, non_blocking=True)
            masks.record_stream(d2h)
            done = torch.cuda.Event()
            done.record()
        return (host, done, save_dir, names)
    enqueue_download.n = 0

    import collections
    jobs = collections.deque()           # downloads in flight, oldest first

    def handle(results):
        for res in results:
            jobs.append(enqueue_download(res))
            while len(jobs) > lag:
                flush(jobs.popleft())

    with ThreadPoolExecutor(max_workers=writer_workers) as writers:
        for vi, video in enumerate(todo):
            frames = data[video]["frames"]
            paths = clip_io.frame_paths(img_folder, video, frames)
            # decode the NEXT videos' JPEGs while this one is on the GPU.  Several ahead (round 6): with launch groups a submit returns
            # at once and the host thread sits in the next replay for a whole group's GPU time -- a video with one expression is
            # 5 ms of GPU against ~8 ms of decode, and with one video of lookahead the decoder idled during every replay
            for ahead in range(1, PREFETCH_VIDEOS + 1):
                if vi + ahead < len(todo):
                    nxt = todo[vi + ahead]
                    cache.prefetch(clip_io.frame_paths(img_folder, nxt, data[nxt]["frames"]))
            for exp_id, item in data[video]["expressions"].items():
                t1 = time.perf_counter()
                clip, orig = cache.get(paths)                                  # decoded / resized once per video
                ids = tokenize(item["exp"]).pin_memory().to(device, non_blocking=True)
                t2 = time.perf_counter()
                save_dir = os.path.join(out_dir, video, exp_id)
                os.makedirs(save_dir, exist_ok=True)
                if streaming:
                    handle(engine.submit(clip, ids, (save_dir, frames), orig))  # results of the PREVIOUS replay, if any
                else:
                    res = engine(clip, ids, orig)                              # [T,H0,W0] bool, still in flight
                    res["tag"] = (save_dir, frames)
                    handle([res])
                stats["seconds_input"] += t2 - t1
                stats["seconds_model"] += time.perf_counter() - t2
                stats["expressions"] += 1
                stats["frames"] += len(frames)
            stats["videos"] += 1
        if streaming:
            handle(engine.drain())
        while jobs:
            flush(jobs.popleft())
        t_tail = time.perf_counter()
        for f in pending:
            f.result()
        stats["seconds_writer_tail"] = time.perf_counter() - t_tail
    torch.cuda.synchronize()
    cache.close()
    stats.update(seconds=time.perf_counter() - t0, cache_hits=cache.hits, cache_misses=cache.misses)
    stats.update({k: v for k, v in getattr(engine, "stats", {}).items()})       # group replays, part-filled ones, remainders
    return stats
