"""Dataset drivers (infer_refytb / infer_davis bodies) on a synthetic dataset directory, against the
reference recipe restated with the CPU oracle, PIL and torch (reference infer_refytb.py:160-277,
infer_davis.py:173-291)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from neurips2023_soc_amd import synthetic_dataset as SD
from neurips2023_soc_amd.infer_davis import davis_palette
from neurips2023_soc_amd.infer_refytb import split_videos

SIZE, MAX_SIZE = 96, 160          # small resize rule so the CPU oracle forward stays in seconds


def test_split_videos_matches_reference_rule():
    vids = [f"v{i}" for i in range(10)]
    parts = [split_videos(vids, r, 3) for r in range(3)]
    assert parts == [vids[0:3], vids[3:6], vids[6:]]          # remainder to the last process
    assert split_videos(vids, 0, 1) == vids


def test_davis_palette_is_the_voc_colormap():
    pal = davis_palette()
    assert len(pal) == 768 and pal[:15] == [0, 0, 0, 128, 0, 0, 0, 128, 0, 128, 128, 0, 0, 0, 128]


def test_hash_tokenizer_shape():
    ids = SD.HashTokenizer()("a dog running left")
    assert ids.shape == (1, 6) and ids[0, 0] == 0 and ids[0, -1] == 2 and int(ids.min()) >= 0
    assert torch.equal(ids, SD.HashTokenizer()("a dog running left"))


def _oracle_clip_outputs(sd, root, video, frames, text, tok, backbone="video-swin-t"):
    """reference recipe on the CPU: PIL resize -> ToTensor -> Normalize -> oracle forward -> selection"""
    from PIL import Image
    from oracle import resize_oracle as R
    from oracle import soc_oracle as O
    mean = torch.tensor([0.485, 0.456, 0.406])[:, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    imgs = []
    for f in frames:
        img = Image.open(os.path.join(root, "valid", "JPEGImages", video, f + ".jpg")).convert("RGB")
        ow0, oh0 = img.size
        oh, ow = R.size_with_aspect_ratio(ow0, oh0, SIZE, MAX_SIZE)
        t = torch.from_numpy(np.array(img.resize((ow, oh), Image.BILINEAR))).permute(2, 0, 1).float().div(255)
        imgs.append(t.sub_(mean).div_(std))
    clip = torch.stack(imgs)
    ids = tok(text)
    out = O.soc_forward(sd, clip, ids, torch.ones_like(ids), clip.shape[-2:], backbone=backbone)
    idx, masks = O.select_query(out)
    return masks, (oh0, ow0)


@pytest.fixture(scope="module")
def gpu_model():
    import neurips2023_soc_amd as S
    from neurips2023_soc_amd import weights as W
    model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
    sd = W.load_synthetic(model, 2023)
    return model.cuda().eval(), sd


@pytest.mark.gpu
@pytest.mark.parametrize("use_graphs", [False, True])
def test_refytb_driver_matches_reference_recipe(gpu_model, tmp_path, use_graphs):
    """use_graphs=True: the driver streams through the software-pipelined hipGraph (results one clip late, drained
    at the end) -- the same PNGs must come out."""
    from PIL import Image
    from neurips2023_soc_amd import infer_refytb
    model, sd = gpu_model
    root = SD.make_dataset(str(tmp_path / "data"), videos=2, frames=3, height=144, width=256, expressions=2, seed=3)
    tok = SD.HashTokenizer()
    out_dir = str(tmp_path / "out")
    stats = infer_refytb.run(model, tok, root, out_dir, size=SIZE, max_size=MAX_SIZE, decode_workers=2,
                             use_graphs=use_graphs)
    assert stats["videos"] == 2 and stats["expressions"] == 4 and stats["frames"] == 12
    assert stats["cache_misses"] == 2 and stats["cache_hits"] == 2      # frames decoded once per video
    _, data = infer_refytb.load_meta(root)
    total = wrong = 0
    for video, item in data.items():
        for exp_id, e in item["expressions"].items():
            logits, (H0, W0) = _oracle_clip_outputs(sd, root, video, item["frames"], e["exp"], tok)
            up = F.interpolate(logits[None], size=(H0, W0), mode="bilinear", align_corners=False)[0]
            want = (up.sigmoid() > 0.5).numpy()
            for j, name in enumerate(item["frames"]):
                png = Image.open(os.path.join(out_dir, video, exp_id, name + ".png"))
                assert png.mode == "L" and png.size == (W0, H0)
                got = np.array(png)
                assert set(np.unique(got)) <= {0, 255}
                bad = (got > 0) != want[j]
                # a flipped pixel must sit on the decision boundary (|logit| at fp32 noise level)
                assert float(up[j].abs()[torch.from_numpy(bad)].max()) < 1e-3 if bad.any() else True
                total += bad.size
                wrong += int(bad.sum())
    assert wrong <= 1e-4 * total


@pytest.mark.gpu
@pytest.mark.parametrize("use_graphs", [False, True])
def test_davis_driver_matches_reference_recipe(gpu_model, tmp_path, use_graphs):
    """use_graphs=True streams the (object, chunk) clips through the software-pipelined replay and merges an
    annotator's label maps when its last result has arrived"""
    from PIL import Image
    from neurips2023_soc_amd import infer_davis, infer_refytb
    model, sd = gpu_model
    # 1 video, 2 objects x 4 annotators = 8 expressions
    root = SD.make_dataset(str(tmp_path / "data"), videos=1, frames=3, height=144, width=256, expressions=8, seed=5)
    tok = SD.HashTokenizer()
    out_dir = str(tmp_path / "out")
    stats = infer_davis.run(model, tok, root, out_dir, size=SIZE, max_size=MAX_SIZE, decode_workers=2,
                            use_graphs=use_graphs)
    # one chunk per video here: the clip is fetched once per annotator (per object in the reference's loop order)
    assert stats["expressions"] == 8 and stats["cache_misses"] == 1 and stats["cache_hits"] == 3
    _, data = infer_refytb.load_meta(root)
    (video, item), = data.items()
    exp_ids = list(item["expressions"].keys())
    total = wrong = 0
    for anno in range(4):
        scores = []
        for obj in range(2):
            text = item["expressions"][exp_ids[obj * 4 + anno]]["exp"]
            logits, (H0, W0) = _oracle_clip_outputs(sd, root, video, item["frames"], text, tok)
            scores.append(F.interpolate(logits[None], size=(H0, W0), mode="bilinear", align_corners=False)[0].sigmoid())
        m = torch.stack(scores)
        m[m < 0.5] = 0.0
        want = torch.cat([torch.full_like(m[:1], 0.1), m], 0).argmax(0).numpy()
        for f in range(3):
            png = Image.open(os.path.join(out_dir, f"anno_{anno}", video, f"{f:05d}.png"))
            assert png.mode == "P" and png.getpalette()[:6] == [0, 0, 0, 128, 0, 0]
            bad = np.array(png) != want[f]
            total += bad.size
            wrong += int(bad.sum())
    assert wrong <= 2e-4 * total


def test_png_writers_formats(tmp_path):
    """8-bit 'L' 0/255 masks (infer_refytb.py:272-277) and palette label maps (infer_davis.py:285-291)."""
    from PIL import Image
    from neurips2023_soc_amd.infer_davis import save_label_map
    from neurips2023_soc_amd.infer_refytb import save_binary_mask
    mask = np.zeros((5, 7), dtype=bool)
    mask[1:3, 2:6] = True
    save_binary_mask(mask, str(tmp_path / "m.png"))
    # the reference's own conversion: float32 mask * 255 -> convert('L')
    ref = Image.fromarray(mask.astype(np.float32) * 255).convert("L")
    got = Image.open(tmp_path / "m.png")
    assert got.mode == "L" and np.array_equal(np.array(got), np.array(ref))
    labels = np.array([[0, 1, 2], [2, 1, 0]], dtype=np.uint8)
    save_label_map(labels, str(tmp_path / "l.png"), davis_palette())
    png = Image.open(tmp_path / "l.png")
    assert png.mode == "P" and np.array_equal(np.array(png), labels)
    assert png.getpalette()[:9] == [0, 0, 0, 128, 0, 0, 0, 128, 0]
