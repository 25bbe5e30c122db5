"""Input side of the inference drivers: JPEG frames -> model input clip.

The reference does, per frame and per expression (infer_refytb.py:193-201, infer_davis.py:214-226):
``Image.open(path).convert('RGB')`` -> ``RandomResize([360], max_size=640)`` (PIL bilinear with
anti-aliasing, datasets/transforms.py:186-216) -> ``ToTensor`` -> ``Normalize`` -> ``torch.stack`` ->
host-to-device copy of the fp32 clip.  Here:

* JPEG decoding stays on the host (PIL, a thread pool -- the decoder releases the GIL) and happens
  once per video, not once per expression;
* the decoded uint8 frames go to the GPU through pinned memory (4x fewer bytes than the fp32 clip at
  720p -> 360p) and K9 (`soc_resize_normalize_u8_f32`) produces the normalised fp32 clip there,
  bit-identical to the CPU pipeline above;
* the resulting device clip is cached per video, so the other expressions of that video reuse it.
"""
from __future__ import annotations

import math
import os
import threading
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hot_ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 32 - 8 - 2


def target_size(width: int, height: int, size: int = 360, max_size: Optional[int] = 640) -> Tuple[int, int]:
    """(out_h, out_w) of the reference's aspect-preserving resize: short side -> `size` unless that
    pushes the long side past `max_size` (datasets/transforms.py:189-207)."""
    w, h = width, height
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def resample_tables(in_size: int, out_size: int):
    """Pillow's bilinear resampling tables for one axis (libImaging/Resample.c: precompute_coeffs with
    the triangle filter, support 1, then normalize_coeffs_8bpc): bounds int32 [out,2] (first source
    index, tap count) and coefficients int32 [out, ksize] with 22 fractional bits.  Plain double
    arithmetic in Pillow's order of operations, so the integers match Pillow's exactly."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = []
        total = 0.0
        for x in range(n):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v)
            total += v
        for x in range(n):
            v = w[x] / total if total != 0.0 else w[x]
            coeffs[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, n)
    return bounds, coeffs


def decode_frame(path: str) -> np.ndarray:
    """JPEG/PNG file -> uint8 [H,W,3] RGB, as Image.open(path).convert('RGB')."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


class PinnedEntry:
    """One pinned staging buffer.  Between `PinnedPool.take()` and `release(event)` it belongs to ONE filler
    (being decoded into / its H2D copy not yet recorded) and is never handed out again."""

    __slots__ = ("tensor", "busy", "in_use")

    def __init__(self, tensor: torch.Tensor):
        self.tensor, self.busy, self.in_use = tensor, None, True

    def release(self, event=None) -> None:
        """The H2D copy out of the buffer has been recorded (`event` fires when it has left the buffer)."""
        self.busy = event
        self.in_use = False

    def __getitem__(self, i):          # entry[0] -> tensor (read-only convenience)
        return (self.tensor, self.busy)[i]


class PinnedPool:
    """A few reusable pinned staging buffers per clip shape.  Pinning fresh host memory per video costs
    ~9 ms for a 22 MB clip and stalls the GPU queue while it happens (measured: +3 ms per clip at three
    expressions per video), so buffers are recycled.  take() is called from the prefetch thread and from the
    main thread (cache miss): the ring is guarded by a lock, an entry stays `in_use` from take() until its
    release(), and a ring whose entries are all in use grows instead of handing one out twice."""

    def __init__(self, depth: int = 3):
        self.depth = depth
        self._bufs: Dict[Tuple[int, ...], List[PinnedEntry]] = {}
        self._turn: Dict[Tuple[int, ...], int] = {}
        self._lock = threading.Lock()

    def take(self, shape: Sequence[int]) -> PinnedEntry:
        """-> an entry owned by the caller until entry.release(); waits until the buffer's previous copy has left it"""
        key = tuple(int(v) for v in shape)
        entry = None
        with self._lock:
            ring = self._bufs.setdefault(key, [])
            if len(ring) >= self.depth:
                turn = self._turn.get(key, 0)
                for step in range(len(ring)):
                    cand = ring[(turn + step) % len(ring)]
                    if not cand.in_use:
                        cand.in_use = True
                        self._turn[key] = (turn + step + 1) % len(ring)
                        entry = cand
                        break
        if entry is None:                          # ring not full yet, or every buffer is with a filler
            buf = torch.empty(key, dtype=torch.uint8)
            entry = PinnedEntry(buf.pin_memory() if torch.cuda.is_available() else buf)
            with self._lock:
                self._bufs[key].append(entry)
            return entry
        if entry.busy is not None:                 # outside the lock: may block on the GPU
            entry.busy.synchronize()
            entry.busy = None
        return entry


def load_frames(paths: Sequence[str], workers: int = 8, pool: Optional[PinnedPool] = None):
    """Decode a clip's frames into one pinned uint8 tensor [T,H0,W0,3] (all frames of a video share a size).
    With a `pool` the staging buffer is recycled and the pool entry is returned instead of the tensor."""
    if not paths:
        raise ValueError("load_frames: empty frame list")
    if workers > 1 and len(paths) > 1:
        with ThreadPoolExecutor(max_workers=min(workers, len(paths))) as threads:
            arrays = list(threads.map(decode_frame, paths))
    else:
        arrays = [decode_frame(p) for p in paths]
    shape = arrays[0].shape
    if any(a.shape != shape for a in arrays):
        raise ValueError("frames of one clip differ in size")
    entry = None
    if pool is not None:
        entry = pool.take((len(arrays), *shape))
        out = entry.tensor
    else:
        out = torch.empty((len(arrays), *shape), dtype=torch.uint8)
        if torch.cuda.is_available():
            out = out.pin_memory()
    view = out.numpy()
    for i, a in enumerate(arrays):
        view[i] = a
    return entry if pool is not None else out


class FramePreprocessor:
    """uint8 frames [T,H0,W0,3] -> (clip [T,3,h,w] float32 on `device`, (H0, W0)).  Tables are cached per
    geometry on the device; the resize + normalisation is K9."""

    def __init__(self, device="cuda", size: int = 360, max_size: Optional[int] = 640,
                 mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD):
        self.device = torch.device(device)
        self.size, self.max_size, self.mean, self.std = size, max_size, tuple(mean), tuple(std)
        self._tables: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}

    def tables(self, in_size: int, out_size: int):
        key = (in_size, out_size)
        if key not in self._tables:
            b, k = resample_tables(in_size, out_size)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
        return self._tables[key]

    def __call__(self, frames: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, int]]:
        T, H0, W0, _ = frames.shape
        h, w = target_size(W0, H0, self.size, self.max_size)
        dev_frames = frames.to(self.device, non_blocking=True)
        clip = hot_ops.resize_normalize(dev_frames, (h, w), self.tables(W0, w), self.tables(H0, h), self.mean, self.std)
        return clip, (H0, W0)


class VideoClipCache:
    """Pre-processed device clips per video (LRU by bytes): the reference re-decodes and re-resizes a
    video's frames for every expression (infer_refytb.py:185-201); a Ref-YouTube-VOS video has ~2-6."""

    def __init__(self, preprocessor: FramePreprocessor, max_bytes: int = 8 << 30, workers: int = 8):
        self.pre, self.max_bytes, self.workers = preprocessor, max_bytes, workers
        self._items: "OrderedDict[Tuple[str, ...], Tuple[torch.Tensor, Tuple[int, int]]]" = OrderedDict()
        self._bytes = 0
        self.hits = self.misses = 0
        self._loader = ThreadPoolExecutor(max_workers=1)     # decodes the NEXT clip while the GPU runs this one
        self._inflight: Dict[Tuple[str, ...], "object"] = {}
        self._pool = PinnedPool()
        self._copy_stream = None

    def close(self) -> None:
        """Stop the background decoder thread (pending prefetches are dropped)."""
        self._loader.shutdown(wait=False, cancel_futures=True)
        self._inflight.clear()

    def _load_and_upload(self, paths):
        """Background thread: decode into pinned memory, then -- on the cache's own copy stream, so the DMA
        and K9 overlap the forward running on the main stream -- upload and pre-process.  Returns the item and
        the event the consumer's stream has to wait for."""
        entry = load_frames(paths, self.workers, self._pool)
        if not (torch.cuda.is_available() and self.pre.device.type == "cuda"):
            item = self.pre(entry.tensor)
            entry.release()
            return item, None
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.pre.device)
        with torch.cuda.stream(self._copy_stream):
            item = self.pre(entry.tensor)
            ready = torch.cuda.Event()
            ready.record()
        entry.release(ready)                 # the staging buffer may be refilled once this copy has left it
        return item, ready

    def prefetch(self, paths: Sequence[str]) -> None:
        """Start decoding / uploading `paths` in the background (no effect on results)."""
        key = tuple(paths)
        if key and key not in self._items and key not in self._inflight:
            self._inflight[key] = self._loader.submit(self._load_and_upload, list(paths))

    def get(self, paths: Sequence[str]) -> Tuple[torch.Tensor, Tuple[int, int]]:
        key = tuple(paths)
        if key in self._items:
            self._items.move_to_end(key)
            self.hits += 1
            return self._items[key]
        self.misses += 1
        fut = self._inflight.pop(key, None)
        if fut is not None:
            item, ready = fut.result()
            if ready is not None:            # produced on the copy stream: order the consumer after it
                torch.cuda.current_stream(self.pre.device).wait_event(ready)
                item[0].record_stream(torch.cuda.current_stream(self.pre.device))
        else:
            entry = load_frames(paths, self.workers, self._pool)
            item = self.pre(entry.tensor)
            done = None
            if item[0].is_cuda:
                done = torch.cuda.Event()
                done.record()
            entry.release(done)
        self._items[key] = item
        self._bytes += item[0].numel() * 4
        while self._bytes > self.max_bytes and len(self._items) > 1:
            _, (old, _) = self._items.popitem(last=False)
            self._bytes -= old.numel() * 4
        return item


def frame_paths(img_folder: str, video: str, frames: Sequence[str], ext: str = ".jpg") -> List[str]:
    return [os.path.join(img_folder, video, f + ext) for f in frames]


class _NoStream:
    """CPU stand-in for a HIP stream: everything is already ordered."""

    def wait_event(self, ev):
        pass


class DoubleBufferedH2D:
    """Host clips -> device through `depth` device slots, copies on their own stream ahead of the compute stream.

    The reference copies every clip to the device inside its loop (`samples.to(device)`, infer_refytb.py:206-212), from
    pageable memory, on the compute stream: the forward waits for 22 MB over PCIe per clip.  Here the host side hands over
    PINNED tensors (PinnedPool, or any `pin_memory()`-ed tensor) and

        feeder.submit(host[0])
        for i in range(n):
            if i + 1 < n: feeder.submit(host[i + 1])      # clip i+1 crosses PCIe while clip i computes
            clip = feeder.acquire()                        # compute stream waits for clip i's copy only
            ... launch the work that reads `clip` ...
            feeder.release()                               # slot may be refilled once that work has run

    Ordering rules (checked by tests/test_clip_io.py with recording stand-ins for streams and events): the copy into a
    slot waits for the `free` event of the work that last read the slot; the compute stream waits for the slot's `ready`
    event; at most `depth` clips are submitted and not yet released, a further submit() raises instead of overwriting.
    `copy_stream` / `compute_stream` / `event_factory` exist for those tests (objects with `wait_event(ev)`; events with
    `record(stream)`); by default they are a fresh HIP stream, the current stream, and `torch.cuda.Event`.
    """

    def __init__(self, shape: Sequence[int], dtype=torch.float32, device="cuda", depth: int = 2,
                 copy_stream=None, compute_stream=None, event_factory=None):
        self.device = torch.device(device)
        self.depth = int(depth)
        if self.depth < 2:
            raise ValueError("one slot cannot be filled while it is being read")
        cuda = self.device.type == "cuda"
        self.slots = [torch.empty(tuple(shape), dtype=dtype, device=self.device) for _ in range(self.depth)]
        if copy_stream is None:
            copy_stream = torch.cuda.Stream(device=self.device) if cuda else _NoStream()
        self._copy = copy_stream
        self._compute = compute_stream                     # None: the current stream at the time of the call
        self._event = event_factory if event_factory is not None else (torch.cuda.Event if cuda else (lambda: None))
        self._ready = [None] * self.depth
        self._free = [None] * self.depth
        self._host = [None] * self.depth                   # keeps the pinned source alive until the slot is released
        self._submitted = self._acquired = self._released = 0

    def _cur(self):
        if self._compute is not None:
            return self._compute
        return torch.cuda.current_stream(self.device) if self.device.type == "cuda" else _NoStream()

    def _record(self, stream):
        ev = self._event()
        if ev is not None:
            ev.record(stream)
        return ev

    def in_flight(self) -> int:
        return self._submitted - self._released

    def submit(self, host: torch.Tensor, on_copied=None) -> int:
        """Enqueue the copy of `host` into the next slot; returns the slot index.  `on_copied(event)` is called with
        the event that fires when the bytes have left `host` (PinnedEntry.release fits)."""
        if self.in_flight() >= self.depth:
            raise RuntimeError("DoubleBufferedH2D: every slot is still in use (submit without a matching release)")
        k = self._submitted % self.depth
        cs = self._copy
        if self._free[k] is not None:
            cs.wait_event(self._free[k])                   # the work that read this slot last has run
        if isinstance(cs, torch.cuda.Stream):
            with torch.cuda.stream(cs):
                self.slots[k].copy_(host.view(self.slots[k].shape), non_blocking=True)
        else:
            self.slots[k].copy_(host.view(self.slots[k].shape))
        ev = self._ready[k] = self._record(cs)
        self._host[k] = host
        self._submitted += 1
        if on_copied is not None:
            on_copied(ev)
        return k

    def acquire(self) -> torch.Tensor:
        """Device tensor of the oldest submitted, not yet acquired clip; the compute stream waits for its copy."""
        if self._acquired >= self._submitted:
            raise RuntimeError("DoubleBufferedH2D: acquire() without a submitted clip")
        if self._acquired > self._released:
            raise RuntimeError("DoubleBufferedH2D: release() the previous clip first")
        k = self._acquired % self.depth
        if self._ready[k] is not None:
            self._cur().wait_event(self._ready[k])
        self._acquired += 1
        return self.slots[k]

    def release(self) -> None:
        """The work reading the acquired slot has been enqueued on the compute stream: mark the point after which the
        slot may be overwritten."""
        if self._released >= self._acquired:
            raise RuntimeError("DoubleBufferedH2D: release() without acquire()")
        k = self._released % self.depth
        self._free[k] = self._record(self._cur())
        self._host[k] = None
        self._released += 1
