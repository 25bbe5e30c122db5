"""Input / output side of the inference drivers: PIL-exact resize + normalise (K9),
DAVIS label merge (K6 DAVIS form), frame loading and caching."""
import numpy as np
import pytest
import torch

from neurips2023_soc_amd import clip_io as CI
from oracle import resize_oracle as R

GEOMETRIES = [((720, 1280), (360, 640)), ((480, 854), (360, 640)), ((100, 37), (20, 640)), ((50, 70), (50, 640)),
              ((33, 45), (64, 640)), ((300, 250), (360, 640)), ((1080, 1920), (360, 640))]


def _frames(T, H0, W0, seed=0):
    rng = np.random.default_rng(seed)
    # smooth-ish content plus noise, full 0..255 range incl. saturated patches
    base = rng.integers(0, 256, (T, H0 // 8 + 1, W0 // 8 + 1, 3)).repeat(8, 1).repeat(8, 2)[:, :H0, :W0]
    noise = rng.integers(-40, 41, (T, H0, W0, 3))
    return np.clip(base + noise, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ CPU: oracle pinned against Pillow
@pytest.mark.parametrize("hw,rule", GEOMETRIES)
def test_resize_oracle_equals_pillow(hw, rule):
    from PIL import Image
    H0, W0 = hw
    img = _frames(1, H0, W0, seed=H0)[0]
    oh, ow = R.size_with_aspect_ratio(W0, H0, *rule)
    want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(R.resize_bilinear_u8(img, oh, ow), want)


def test_preprocess_oracle_equals_torch_ops():
    """ToTensor + Normalize as torchvision does them on the CPU: .div(255), .sub_(mean).div_(std)."""
    frames = _frames(2, 96, 128)
    clip, small = R.preprocess_clip(frames, 48, 640)
    t = torch.from_numpy(small).permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
    mean = torch.as_tensor(CI.IMAGENET_MEAN, dtype=torch.float32)
    std = torch.as_tensor(CI.IMAGENET_STD, dtype=torch.float32)
    t.sub_(mean[None, :, None, None]).div_(std[None, :, None, None])
    assert np.array_equal(clip, t.numpy())


@pytest.mark.parametrize("hw,rule", GEOMETRIES)
def test_product_tables_equal_oracle(hw, rule):
    H0, W0 = hw
    oh, ow = CI.target_size(W0, H0, *rule)
    assert (oh, ow) == R.size_with_aspect_ratio(W0, H0, *rule)
    for a, b in ((W0, ow), (H0, oh)):
        xmin, n, k, ksize = R.coeffs_1d(a, b)
        bounds, coeffs = CI.resample_tables(a, b)
        assert coeffs.shape == (b, ksize) and coeffs.dtype == np.int32
        assert np.array_equal(bounds[:, 0], xmin) and np.array_equal(bounds[:, 1], n)
        assert np.array_equal(coeffs, k)


def test_target_size_rule():
    assert CI.target_size(1280, 720) == (360, 640)      # the Ref-YouTube-VOS case
    assert CI.target_size(854, 480) == (360, 640)       # DAVIS 480p: long side capped at 640 -> 359.7 -> 360
    assert CI.target_size(720, 1280) == (640, 360)      # portrait
    assert CI.target_size(640, 360) == (360, 640)       # already there
    assert CI.target_size(500, 500, 360, 640) == (360, 360)


def test_load_frames_and_cache(tmp_path):
    from PIL import Image
    frames = _frames(3, 40, 56)
    paths = []
    for i, f in enumerate(frames):
        p = tmp_path / f"{i:05d}.png"                   # PNG: lossless, so the round trip is exact
        Image.fromarray(f).save(p)
        paths.append(str(p))
    got = CI.load_frames(paths, workers=2)
    assert got.dtype == torch.uint8 and np.array_equal(got.numpy(), frames)
    with pytest.raises(ValueError):
        CI.load_frames([])
    calls = []

    class Pre:
        def __call__(self, fr):
            calls.append(fr.shape)
            return torch.zeros(fr.shape[0], 3, 4, 4), (40, 56)
    cache = CI.VideoClipCache(Pre(), workers=1)
    a = cache.get(paths)
    b = cache.get(paths)
    assert a[0] is b[0] and len(calls) == 1 and (cache.hits, cache.misses) == (1, 1)
    cache.prefetch(paths[:2])                                # background decode, consumed by the next get
    cache.prefetch(paths[:2])
    assert len(cache._inflight) == 1
    c = cache.get(paths[:2])
    assert calls[-1] == (2, 40, 56, 3) and not cache._inflight and c[1] == (40, 56)


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ops():
    from neurips2023_soc_amd import build_ext, hot_ops
    build_ext.build(verbose=False)
    return hot_ops


@pytest.mark.gpu
@pytest.mark.parametrize("hw,rule,T", [(g[0], g[1], 2) for g in GEOMETRIES] + [((720, 1280), (360, 640), 8)])
def test_resize_normalize_bit_exact(ops, hw, rule, T):
    H0, W0 = hw
    frames = _frames(T, H0, W0, seed=W0)
    want, want_u8 = R.preprocess_clip(frames, *rule)
    pre = CI.FramePreprocessor("cuda", *rule)
    h, w = CI.target_size(W0, H0, *rule)
    got, got_u8 = ops.resize_normalize(torch.from_numpy(frames).cuda(), (h, w), pre.tables(W0, w), pre.tables(H0, h),
                                       CI.IMAGENET_MEAN, CI.IMAGENET_STD, return_u8=True)
    assert np.array_equal(got_u8.cpu().numpy(), want_u8)          # PIL's bytes
    assert np.array_equal(got.cpu().numpy(), want)                # and the same fp32 bits after normalisation
    clip, orig = pre(torch.from_numpy(frames))
    assert orig == (H0, W0) and torch.equal(clip, got)


@pytest.mark.gpu
def test_frame_pipeline_from_files(ops, tmp_path):
    """JPEG files -> device clip, against the reference recipe run with PIL + torch on the CPU."""
    from PIL import Image
    frames = _frames(4, 144, 256)
    paths = []
    for i, f in enumerate(frames):
        p = tmp_path / f"{i:05d}.jpg"
        Image.fromarray(f).save(p, quality=90)
        paths.append(str(p))
    mean = torch.as_tensor(CI.IMAGENET_MEAN)[:, None, None]
    std = torch.as_tensor(CI.IMAGENET_STD)[:, None, None]
    want = []
    for p in paths:                                                # infer_refytb.py:193-201
        img = Image.open(p).convert("RGB")
        oh, ow = R.size_with_aspect_ratio(*img.size, 72, 128)
        img = img.resize((ow, oh), Image.BILINEAR)
        t = torch.from_numpy(np.array(img)).permute(2, 0, 1).contiguous().float().div(255)
        want.append(t.sub_(mean).div_(std))
    cache = CI.VideoClipCache(CI.FramePreprocessor("cuda", 72, 128), workers=2)
    clip, orig = cache.get(paths)
    assert orig == (144, 256) and torch.equal(clip.cpu(), torch.stack(want))
    assert cache.get(paths)[0] is clip


@pytest.mark.gpu
@pytest.mark.parametrize("O,T,h,w,H0,W0", [(3, 4, 90, 160, 480, 854), (1, 2, 9, 11, 31, 50), (5, 1, 30, 40, 30, 40)])
def test_upsample_merge_labels_vs_torch(ops, O, T, h, w, H0, W0):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(O + T)
    logits = torch.randn(O, T, h, w, generator=g) * 4
    logits[:, :, : h // 2] += 30.0                                # saturated sigmoids: exact ties between objects
    up = F.interpolate(logits, size=(H0, W0), mode="bilinear", align_corners=False).sigmoid()
    m = up.clone()
    m[m < 0.5] = 0.0                                               # infer_davis.py:264-268
    want = torch.cat([torch.full((1, T, H0, W0), 0.1), m], 0).argmax(0)
    got = ops.upsample_merge_labels(logits.cuda(), (H0, W0)).cpu().long()
    bad = got != want
    if bad.any():   # only where the two best scores are closer than the fp32 noise of interpolate + exp
        top2 = torch.cat([torch.full((1, T, H0, W0), 0.1), m], 0).topk(2, 0)[0]
        assert float((top2[0] - top2[1])[bad].max()) < 1e-6 or float((up - 0.5).abs().min(0)[0][bad].max()) < 1e-6
    assert float(bad.float().mean()) < 1e-4


def test_pinned_pool_never_hands_out_a_buffer_twice():
    """take() runs on the prefetch thread and on the main thread (cache miss): an entry stays with its filler
    until release(); two concurrent fillers of the same shape must get different buffers, and a ring whose
    entries are all in use grows."""
    import threading
    pool = CI.PinnedPool(depth=2)
    shape = (2, 4, 4, 3)
    held, lock, errors = [], threading.Lock(), []

    def filler(tag):
        for _ in range(200):
            e = pool.take(shape)
            with lock:
                if any(e is h for h in held):
                    errors.append("entry handed out twice")
                held.append(e)
            e.tensor.fill_(tag)
            if int(e.tensor.max()) != tag or int(e.tensor.min()) != tag:
                errors.append("buffer overwritten while in use")
            with lock:
                held.remove(e)
            e.release()

    threads = [threading.Thread(target=filler, args=(t,)) for t in (1, 2, 3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    ring = pool._bufs[tuple(shape)]
    assert 2 <= len(ring) <= 3 and not any(e.in_use for e in ring)
    a, b = pool.take(shape), pool.take(shape)
    c = pool.take(shape)                      # both ring entries (or all three) busy -> distinct third / fourth
    assert len({id(a), id(b), id(c)}) == 3


def test_double_buffered_h2d_ordering():
    """The streamed form of the reference's per-clip `samples.to(device)` (infer_refytb.py:206-212; SURVEY 8e: pinned,
    double-buffered H2D).  Streams and events are recording stand-ins, so the test sees the ORDER of operations the
    feeder would enqueue: the copy into a slot waits for the release of the work that read it last, the compute stream
    waits for exactly the copy of the clip it is about to read, and a third un-released submit raises."""
    log = []

    class Ev:
        n = 0

        def __init__(self):
            Ev.n += 1
            self.id, self.on = Ev.n, None

        def record(self, stream):
            self.on = stream.name
            log.append(("record", stream.name, self.id))

    class St:
        def __init__(self, name):
            self.name = name

        def wait_event(self, ev):
            log.append(("wait", self.name, ev.id, ev.on))

    n = 7
    hosts = [torch.full((2, 3), float(i)) for i in range(n)]
    f = CI.DoubleBufferedH2D((2, 3), device="cpu", depth=2, copy_stream=St("copy"), compute_stream=St("compute"),
                             event_factory=Ev)
    seen, copied = [], []
    f.submit(hosts[0], on_copied=copied.append)
    for i in range(n):
        if i + 1 < n:
            f.submit(hosts[i + 1], on_copied=copied.append)
            with pytest.raises(RuntimeError):
                f.submit(hosts[0])               # both slots hold clips that have not been released
        clip = f.acquire()
        seen.append(float(clip[0, 0]))           # "launch the work that reads clip"
        assert float(clip.min()) == float(clip.max()) == i
        with pytest.raises(RuntimeError):
            f.acquire()                          # one clip at a time on the compute stream
        f.release()
    assert seen == [float(i) for i in range(n)] and f.in_flight() == 0 and len(copied) == n
    with pytest.raises(RuntimeError):
        f.acquire()
    with pytest.raises(RuntimeError):
        f.release()
    # per clip i >= 2 the copy stream waited, before recording the copy's `ready`, for the event the compute stream
    # recorded when it released clip i-2 (same slot); the compute stream waited for clip i's own `ready`
    ready = [e for e in log if e[0] == "record" and e[1] == "copy"]
    free = [e for e in log if e[0] == "record" and e[1] == "compute"]
    cwait = [e for e in log if e[0] == "wait" and e[1] == "copy"]
    kwait = [e for e in log if e[0] == "wait" and e[1] == "compute"]
    assert len(ready) == n and len(free) == n and len(cwait) == n - 2 and len(kwait) == n
    for i in range(n):
        assert kwait[i][2] == ready[i][2] and kwait[i][3] == "copy"
        assert log.index(ready[i]) < log.index(kwait[i])
    for i in range(2, n):
        assert cwait[i - 2][2] == free[i - 2][2] and cwait[i - 2][3] == "compute"
        assert log.index(free[i - 2]) < log.index(cwait[i - 2]) < log.index(ready[i])
    with pytest.raises(ValueError):
        CI.DoubleBufferedH2D((2, 3), device="cpu", depth=1)
