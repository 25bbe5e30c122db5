"""The kernel every pixel-sized linear layer takes, pinned per BASELINE configuration.

CPU: `routes.table()` (stand-in tensors through the real dispatch predicates) against tests/golden/routes.json -- a threshold
change that silently sends a layer back to the library GEMM fails here and has to be acknowledged by re-generating the golden
(`python -m neurips2023_soc_amd.routes --write`).  GPU: the launches a real forward records are the ones the table names."""
import json

import pytest
import torch

from neurips2023_soc_amd import routes


def test_route_table_matches_the_golden():
    with open(routes.GOLDEN) as f:
        want = json.load(f)
    got = routes.all_tables()
    assert set(got) == set(want)
    for cfg in want:
        assert got[cfg] == want[cfg], (cfg, {k: (got[cfg].get(k), want[cfg].get(k))
                                             for k in set(got[cfg]) | set(want[cfg]) if got[cfg].get(k) != want[cfg].get(k)})


def test_route_table_needs_no_native_library(monkeypatch):
    """Routing reads shapes, dtypes and devices only: on a checkout without libsoc_hip.so (it is git-ignored) and without hipcc the
    table still comes out (mlp_split_supported used to load the library for soc_mlp_split_max_hidden)."""
    from neurips2023_soc_amd import _lib

    def no_library(*a, **k):
        raise _lib.SocHipError("libsoc_hip.so is missing (test)")
    monkeypatch.setattr(_lib, "load", no_library)
    with open(routes.GOLDEN) as f:
        want = json.load(f)
    assert routes.table("video-swin-t", 8, 360, 640) == want["video-swin-t T=8 360x640"]
    assert routes.table("video-swin-t", 8, 360, 640, 10) == want["video-swin-t T=8 360x640 x10 clips"]


def test_headline_config_keeps_its_hand_written_kernels():
    """The BASELINE headline (Swin-T, T = 8, 360 x 640): every MLP of stages 0-2 and the encoder's feed-forward block on K23,
    qkv / proj of stages 0-1 on K13b, of stages 2-3 and the encoder's value / output projections on K24; what is left to the library is the
    list below (hidden width 3072 of stage 3, the K = 1536 reduction, the coarse levels' narrow projections)."""
    t = routes.table("video-swin-t", 8, 360, 640)
    assert [t[f"swin{s}.mlp"] for s in range(3)] == ["k23"] * 3 and t["encoder.ffn"] == "k23"
    assert all(t[f"swin{s}.{n}"] == "k13b" for s in range(2) for n in ("qkv", "proj"))
    assert all(t[f"swin{s}.{n}"] == "k24" for s in (2, 3) for n in ("qkv", "proj")) and t["swin3.fc1"] == "k24"
    assert t["encoder.value_proj"] == t["encoder.output_proj"] == "k24" and t["encoder.offsets|weights"] == "k20"
    library = sorted(k for k, v in t.items() if v == "library")
    assert library == ["input_proj3", "merge2", "swin3.fc2", "vlf2.out*tgt", "vlf2.q", "vlf3.out*tgt", "vlf3.q"], library


def test_f32_mode_sends_nothing_to_the_bf16_kernels():
    from neurips2023_soc_amd import hot_ops
    with hot_ops.use_matmul_mode("f32"):
        t = routes.table("video-swin-t", 8, 360, 640)
    assert hot_ops.matmul_mode() == "split"
    assert all(v not in ("k23", "k20", "k24") for v in t.values()), t  # ("k13b" sites run the f32-MFMA form of K13 then)


@pytest.mark.gpu
def test_recorded_launches_are_the_ones_the_table_names():
    """One eager forward of the headline configuration with the call recorders on: every site the table sends to K13b / K20 /
    K23 shows up in that kernel's recorded calls with its (rows, N, K)."""
    import neurips2023_soc_amd as S
    from neurips2023_soc_amd import hot_ops, weights as W
    T, H, Wd, L = 8, 360, 640, 10
    dev = torch.device("cuda")
    model, _, _ = S.build_model(S.default_args("video-swin-t", text_encoder_random_init=True))
    W.load_synthetic(model, 2023)
    model = model.to(dev).eval()
    clip = W.synthetic_clip(1, T, H, Wd).to(dev)
    ids = W.synthetic_token_ids(1, L)
    text = {"input_ids": ids.to(dev), "attention_mask": torch.ones_like(ids).to(dev)}
    pad = torch.zeros(T, 1, H, Wd, dtype=torch.bool, device=dev)
    targets = [[{"size": (H, Wd)}] for _ in range(T)]
    fresh = lambda: S.NestedTensor(clip[:, None], pad, unpadded=True)     # noqa: E731  (the forward rewrites its layout in place)
    model(fresh(), None, text, targets)                     # warm-up (weight images, caches)
    for rec in (hot_ops.record_ws_linear_calls, hot_ops.record_linear_split_calls, hot_ops.record_mlp_split_calls,
                hot_ops.record_xs_linear_calls):
        rec(True)
    model(fresh(), None, text, targets)
    k13 = {(c["x"].numel() // c["weight"].shape[1], *c["weight"].shape) for c in hot_ops.record_ws_linear_calls(False)}
    k20 = {(c["x"].numel() // c["weight"].shape[1], *c["weight"].shape) for c in hot_ops.record_linear_split_calls(False)}
    k23 = {(c["x"].numel() // c["w1"].shape[1], c["w1"].shape[1], c["w1"].shape[0]) for c in hot_ops.record_mlp_split_calls(False)}
    k24 = {(c["x"].numel() // c["weight"].shape[1], *c["weight"].shape) for c in hot_ops.record_xs_linear_calls(False)}
    t = routes.table("video-swin-t", T, H, Wd)
    rows = [T * 90 * 160, T * 45 * 80, T * 23 * 40, T * 12 * 20]
    for s, C in enumerate((96, 192, 384)):
        assert (rows[s], C, 4 * C) in k23, (s, sorted(k23))
    for s, C in enumerate((96, 192)):
        assert (rows[s], 3 * C, C) in k13 and (rows[s], C, C) in k13, (s, sorted(k13))
    for s, C in ((2, 384), (3, 768)):
        assert (rows[s], 3 * C, C) in k24 and (rows[s], C, C) in k24, (s, sorted(k24))
    enc = T * (45 * 80 + 23 * 40 + 12 * 20 + 6 * 10)
    assert (enc, 256, 2048) in k23 and (enc, 256, 256) in k24 and (enc, 384, 256) in k20      # FFN, value / output proj, offsets | weights
    assert t["encoder.value_proj"] == t["encoder.output_proj"] == "k24"
    assert (rows[1], 192, 384) in k24 and (rows[2], 384, 768) in k24      # merge0, merge1
    assert t["swin3.fc1"] == "k24" and (rows[3], 3072, 768) in k24
