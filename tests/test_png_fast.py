"""libsoc_host.so (include/soc_host.h): the run-length PNG encoder of the drivers' output side, against Pillow as the
decoder (it verifies the zlib Adler-32 and every chunk CRC) and as the reference writer (infer_refytb.py:269-277,
infer_davis.py:285-291: same mode, size, pixels, palette)."""
import io
import os
import re
import threading

import numpy as np
import pytest
from PIL import Image

from neurips2023_soc_amd import png_fast
from neurips2023_soc_amd.infer_davis import davis_palette, save_label_map
from neurips2023_soc_amd.infer_refytb import save_binary_mask

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def decode(data):
    img = Image.open(io.BytesIO(data))
    img.load()
    return img


def test_library_exports_every_declared_symbol():
    lib = png_fast.load()
    hdr = open(os.path.join(ROOT, "include", "soc_host.h")).read()
    declared = set(re.findall(r"\b(soc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(png_fast.EXPORTS)
    assert all(hasattr(lib, n) for n in declared) and lib.soc_host_abi_version() == png_fast.ABI_VERSION == 1
    assert lib.soc_png_bound(0, 5) == 0 and lib.soc_png_bound(720, 1280) > 720 * 1281


@pytest.mark.parametrize("h,w", [(1, 1), (1, 7), (3, 5), (7, 300), (64, 64), (361, 641), (720, 1280)])
def test_round_trip_of_masks_labels_and_noise(h, w):
    rng = np.random.default_rng(h * 1000 + w)
    yy, xx = np.mgrid[:h, :w]
    mask = ((yy - h / 3) ** 2 + (xx - w / 2) ** 2 < (min(h, w) / 3) ** 2) | ((yy > 0.7 * h) & (xx % 97 < 40))
    img = decode(png_fast.encode(mask, binarize=True))
    assert img.mode == "L" and img.size == (w, h) and np.array_equal(np.array(img), mask.astype(np.uint8) * 255)
    for const in (0, 1, 255):                                  # one run over the whole image, incl. the 258-byte run cap
        a = np.full((h, w), const, np.uint8)
        assert np.array_equal(np.array(decode(png_fast.encode(a))), a)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)        # worst case: no runs, every literal
    assert np.array_equal(np.array(decode(png_fast.encode(noise))), noise)
    labels = (rng.integers(1, 6, (h, w)) * (rng.random((h, w)) < 0.02)).astype(np.uint8)
    pal = davis_palette()
    img = decode(png_fast.encode(labels, palette=pal))
    assert img.mode == "P" and np.array_equal(np.array(img), labels) and img.getpalette()[:768] == list(pal)
    wide = rng.integers(0, 2, (h, w + 9), dtype=np.uint8)       # a view with a row stride
    view = wide[:, 4:4 + w]
    assert np.array_equal(np.array(decode(png_fast.encode(view, binarize=True))), view * 255)
    runs = np.repeat(rng.integers(0, 256, (h, -(-w // 3)), dtype=np.uint8), 3, axis=1)[:, :w]      # runs of exactly 3 (shortest match)
    assert np.array_equal(np.array(decode(png_fast.encode(runs))), runs)


def test_writers_match_pillow_pixel_for_pixel(tmp_path, monkeypatch):
    rng = np.random.default_rng(7)
    mask = rng.random((90, 160)) < 0.4
    mask = np.kron(mask, np.ones((8, 8), dtype=bool))           # 720 x 1280, blocky
    labels = (np.kron(rng.integers(0, 4, (45, 80)), np.ones((16, 16), dtype=np.int64))).astype(np.uint8)
    pal = davis_palette()
    save_binary_mask(mask, str(tmp_path / "a.png"))
    save_label_map(labels, str(tmp_path / "b.png"), pal)
    monkeypatch.setenv("SOC_PNG", "pillow")                     # the reference's writer
    save_binary_mask(mask, str(tmp_path / "a_ref.png"))
    save_label_map(labels, str(tmp_path / "b_ref.png"), pal)
    for name in ("a", "b"):
        got, ref = Image.open(tmp_path / f"{name}.png"), Image.open(tmp_path / f"{name}_ref.png")
        assert got.mode == ref.mode and got.size == ref.size and np.array_equal(np.array(got), np.array(ref))
        assert got.getpalette() == ref.getpalette()
    assert os.path.getsize(tmp_path / "a.png") < 40_000        # piecewise constant: a few KB, not 900 KB
    # Pillow at its default level is what the reference writes
    monkeypatch.setenv("SOC_PNG_LEVEL", "1")
    save_binary_mask(mask, str(tmp_path / "a_l1.png"))
    assert os.path.getsize(tmp_path / "a_l1.png") >= os.path.getsize(tmp_path / "a_ref.png")


def test_bad_arguments_and_small_buffers():
    lib = png_fast.load()
    a = np.zeros((4, 4), np.uint8)
    out = np.zeros(16, np.uint8)
    assert lib.soc_png_encode_u8(a.ctypes.data, 4, 4, 4, 0, None, 0, out.ctypes.data, out.size) == -2      # capacity
    big = np.zeros(lib.soc_png_bound(4, 4), np.uint8)
    assert lib.soc_png_encode_u8(None, 4, 4, 4, 0, None, 0, big.ctypes.data, big.size) == -1
    assert lib.soc_png_encode_u8(a.ctypes.data, 4, 4, 3, 0, None, 0, big.ctypes.data, big.size) == -1      # stride < width
    assert lib.soc_png_encode_u8(a.ctypes.data, 4, 4, 4, 0, a.ctypes.data, 0, big.ctypes.data, big.size) == -1   # empty palette
    with pytest.raises(ValueError):
        png_fast.encode(np.zeros((4, 4), np.float32))


def test_thread_safe_from_a_writer_pool():
    """The drivers call the encoder from 16 writer threads: no shared state, one scratch buffer per thread."""
    rng = np.random.default_rng(3)
    masks = [np.kron(rng.random((30, 40)) < 0.5, np.ones((6, 8), dtype=bool)) for _ in range(8)]
    want = [png_fast.encode(m, binarize=True) for m in masks]
    errors = []

    def worker(i):
        for _ in range(40):
            if png_fast.encode(masks[i], binarize=True) != want[i]:
                errors.append(i)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


def test_missing_host_library_falls_back_to_pillow(monkeypatch):
    """No gcc / a read-only package directory: the drivers are told to write with Pillow (one warning) BEFORE any forward runs,
    instead of raising inside the writer pool at the end of the run."""
    import warnings
    from neurips2023_soc_amd import png_fast

    def no_gcc(*a, **k):
        raise FileNotFoundError("gcc")
    monkeypatch.setattr(png_fast, "_fallback", None)
    monkeypatch.setattr(png_fast, "load", no_gcc)
    monkeypatch.delenv("SOC_PNG", raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert png_fast.use_pillow() is True and png_fast.use_pillow() is True
    assert len(w) == 1 and "Pillow" in str(w[0].message)
    monkeypatch.setattr(png_fast, "_fallback", None)
