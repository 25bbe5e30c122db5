"""Ref-DAVIS17 inference driver: the per-process body of the reference's infer_davis.py
(`sub_processor`, :124-297) on the MI355X pipeline.

    for each video, for each of the 4 annotators                        reference :173-196
      for each object: the video in clips of <= 36 frames              :199-216  (MSDA im2col_step limit)
        frames -> clip (cached per video chunk), forward, best query   :217-246
      merge the objects: up-sample, sigmoid, zero < 0.5, background 0.1, argmax   :248-272   one K6 launch
      write <out>/anno_<a>/<video>/<%05d>.png as palette PNG            :285-291

Expression i belongs to object i // 4 and annotator i % 4 (:193-199).
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import clip_io, hot_ops
from .infer import ClipInferencer
from .infer_refytb import Tokenize, load_meta, split_videos

CLIP_LEN = 36


def davis_palette() -> list:
    """The PASCAL-VOC / DAVIS colour map (bit-interleaved label -> RGB), what the reference reads from
    valid/Annotations/blackswan/00000.png (:168-169)."""
    pal = []
    for i in range(256):
        r = g = b = 0
        c = i
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal += [r, g, b]
    return pal


def save_label_map(labels: np.ndarray, path: str, palette: Sequence[int]) -> None:
    """[H,W] object labels -> 8-bit palette PNG (reference infer_davis.py:285-291): same pixels and palette; written by the
    run-length encoder of libsoc_host.so unless SOC_PNG=pillow (see infer_refytb.save_binary_mask)."""
    from . import png_fast
    if png_fast.use_pillow():
        from PIL import Image
        img = Image.fromarray(labels.astype(np.uint8))
        img.putpalette(list(palette))
        img.save(path, **png_fast.pillow_save_kwargs())
    else:
        png_fast.save(path, labels.astype(np.uint8, copy=False), palette=palette)


@torch.no_grad()
def run(model, tokenize: Tokenize, root: str, out_dir: str, rank: int = 0, world: int = 1, device="cuda",
        split: str = "valid", size: int = 360, max_size: Optional[int] = 640, use_graphs: bool = False,
        decode_workers: int = 8, writer_workers: int = 16, palette: Optional[Sequence[int]] = None,
        videos: Optional[Sequence[str]] = None, annotators: int = 4,
        engine: Optional[ClipInferencer] = None, pad_tokens_to: Optional[int] = 32, group: int = 1) -> Dict:
    img_folder, data = load_meta(root, split)
    if palette is None:
        ref_png = os.path.join(root, split, "Annotations", "blackswan", "00000.png")
        if os.path.exists(ref_png):
            from PIL import Image
            palette = Image.open(ref_png).getpalette()
        else:
            palette = davis_palette()
    todo = split_videos(sorted(data.keys()) if videos is None else list(videos), rank, world)
    engine = engine or ClipInferencer(model, device, use_graphs=use_graphs, pad_tokens_to=pad_tokens_to, group=group)
    cache = clip_io.VideoClipCache(clip_io.FramePreprocessor(device, size, max_size), workers=decode_workers)
    stats = {"videos": 0, "expressions": 0, "frames": 0, "seconds_input": 0.0, "seconds_model": 0.0}
    pending = []
    t0 = time.perf_counter()
    # With graphs the clips stream through the software-pipelined replay (ClipInferencer.submit): a result arrives one
    # submit late, so an annotator's label maps are merged once all its (object, chunk) results are in -- while the next
    # annotator's clips already run.  Chunks are walked outermost so that clips of one length follow each other (a length
    # change drains the pipeline); per (object, chunk) the forward is the same as in the object-major order of the
    # reference (:199-246).
    streaming = engine.use_graphs
    waiting = []                # annotator jobs whose results are still arriving, oldest first

    def place(res):
        job, obj, ci = res["tag"]
        job["got"][obj][ci] = res["mask_logits"].clone()        # a view of the graph's record: copy before the next submit
        job["left"] -= 1

    def merge_ready(writers, everything=False):
        while waiting and (waiting[0]["left"] == 0):
            job = waiting.pop(0)
            t2 = time.perf_counter()
            per_obj = torch.stack([torch.cat(chunks, 0) for chunks in job["got"]])
            labels = hot_ops.upsample_merge_labels(per_obj, job["orig"]).cpu().numpy()   # [T,H0,W0]
            stats["seconds_model"] += time.perf_counter() - t2    # waits for this annotator's forwards
            os.makedirs(job["save_dir"], exist_ok=True)
            for f in range(labels.shape[0]):
                pending.append(writers.submit(save_label_map, labels[f], os.path.join(job["save_dir"], f"{f:05d}.png"),
                                              palette))
        assert not (everything and waiting), "results missing after the pipeline was drained"

    with ThreadPoolExecutor(max_workers=writer_workers) as writers:
        for vi, video in enumerate(todo):
            frames = data[video]["frames"]
            exps = data[video]["expressions"]
            if vi + 1 < len(todo):      # decode the next video's first clip while this one is on the GPU
                nxt = todo[vi + 1]
                cache.prefetch(clip_io.frame_paths(img_folder, nxt, data[nxt]["frames"][:CLIP_LEN]))
            exp_ids = list(exps.keys())
            num_obj = len(exp_ids) // annotators
            starts = list(range(0, len(frames), CLIP_LEN))
            for anno in range(annotators):
                ids = [tokenize(exps[exp_ids[obj * annotators + anno]]["exp"]).to(device) for obj in range(num_obj)]
                job = {"got": [[None] * len(starts) for _ in range(num_obj)], "left": num_obj * len(starts), "orig": None,
                       "save_dir": os.path.join(out_dir, f"anno_{anno}", video)}
                waiting.append(job)
                for ci, c0 in enumerate(starts):
                    t1 = time.perf_counter()
                    clip, job["orig"] = cache.get(clip_io.frame_paths(img_folder, video, frames[c0:c0 + CLIP_LEN]))
                    stats["seconds_input"] += time.perf_counter() - t1
                    for obj in range(num_obj):
                        if streaming:
                            for res in engine.submit(clip, ids[obj], (job, obj, ci)):   # results of the PREVIOUS replay, if any
                                place(res)
                        else:
                            job["got"][obj][ci] = engine(clip, ids[obj])["mask_logits"].clone()   # [t,h,w]
                            job["left"] -= 1
                    merge_ready(writers)
                stats["expressions"] += num_obj
                stats["frames"] += num_obj * len(frames)
            stats["videos"] += 1
        if streaming:
            for res in engine.drain():
                place(res)
        merge_ready(writers, everything=True)
        t_tail = time.perf_counter()
        for f in pending:
            f.result()
        stats["seconds_writer_tail"] = time.perf_counter() - t_tail
    torch.cuda.synchronize()
    cache.close()
    stats.update(seconds=time.perf_counter() - t0, cache_hits=cache.hits, cache_misses=cache.misses)
    stats.update({k: v for k, v in getattr(engine, "stats", {}).items()})       # group replays, part-filled ones, remainders
    return stats
