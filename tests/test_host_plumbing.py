"""CPU (-m "not gpu") checks of the host side of the product package.

The product path has NO CPU fallback (test_hot_ops_refuse_cpu).  To still exercise the Python
plumbing around the kernels here, the four hot-op entry points are monkeypatched -- in this test
only -- with the oracle's kernel-boundary functions; the result must reproduce the reference's
golden outputs, which proves module wiring, layouts and state_dict naming independently of HIP.
"""
import os

import numpy as np
import pytest
import torch

import neurips2023_soc_amd as S
from neurips2023_soc_amd import hot_ops, weights as W
from oracle import soc_oracle as O
from tests.golden_utils import t


def maxdiff(a, b):
    return float((torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


@pytest.fixture(scope="module")
def cpu_model(synthetic_sd):
    model, criterion, post = S.build_model(S.default_args(text_encoder_random_init=True, device="cpu"))
    assert criterion is None and post is not None
    missing, unexpected = model.load_state_dict(synthetic_sd, strict=False)
    assert not unexpected
    assert all(k.endswith(("relative_position_index", "position_ids", "token_type_ids")) for k in missing)
    return model.eval()


@pytest.mark.parametrize("tag,backbone", [("t", "video-swin-t"), ("s", "video-swin-s"), ("b", "video-swin-b")])
def test_state_dict_matches_reference_checkpoint_layout(ref_shapes, tag, backbone):
    """Every key / shape / dtype of the reference state_dict exists here (SURVEY 8b checkpoint API)."""
    model, _, _ = S.build_model(S.default_args(backbone, text_encoder_random_init=True))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
    assert ours == ref_shapes(tag)


def test_bbox_embed_is_shared_with_decoder(cpu_model):
    assert cpu_model.transformer.decoder.bbox_embed is cpu_model.bbox_embed


def test_hot_ops_refuse_cpu(cpu_model):
    x = torch.zeros(1, 2, 7, 7, 96)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hot_ops.window_attention3d(x, torch.zeros(96), torch.zeros(2535, 1), 1, (8, 7, 7), (0, 0, 0))
    clip = W.synthetic_clip(1, 2, 64, 64)
    samples = S.nested_tensor_from_videos_list([clip])
    ids = W.synthetic_token_ids(1, 5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cpu_model(samples, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)},
                  [[{"size": torch.tensor([64, 64])}]] * 2)


def test_strings_without_tokenizer_raise(cpu_model):
    with pytest.raises(RuntimeError, match="pre-tokenised"):
        cpu_model.forward_text(["a cat"], torch.device("cpu"))


def test_train_mode_refused(cpu_model):
    cpu_model.train()
    try:
        with pytest.raises(RuntimeError, match="inference-only"):
            cpu_model(None, None, None, None)
    finally:
        cpu_model.eval()


@pytest.fixture()
def oracle_kernels(monkeypatch):
    monkeypatch.setattr(hot_ops, "msda_forward", O.msda_core)
    monkeypatch.setattr(hot_ops, "window_attention3d", O.window_attention_core)
    monkeypatch.setattr(hot_ops, "mha_core", O.mha_core)
    monkeypatch.setattr(hot_ops, "dynamic_mask", O.dynamic_mask_core)
    monkeypatch.setattr(hot_ops, "add_layernorm", O.add_layernorm_core)
    monkeypatch.setattr(hot_ops, "box_refine", O.box_refine_core)
    monkeypatch.setattr(hot_ops, "patch_merge_layernorm", O.patch_merge_layernorm_core)


def run_cfg(model, g):
    seed, T, H, Wd, L = (int(v) for v in g["cfg"])
    samples = S.nested_tensor_from_videos_list([W.synthetic_clip(seed, T, H, Wd)])
    ids = W.synthetic_token_ids(seed, L)
    targets = [[{"size": torch.tensor([H, Wd])}] for _ in range(T)]
    return model(samples, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)}, targets)


def test_plumbing_reproduces_reference_tiny(cpu_model, oracle_kernels, golden):
    g = golden("tiny_forward.npz")
    out = run_cfg(cpu_model, g)
    assert set(out) == {"pred_masks", "pred_logit", "pred_boxes", "text_sentence_feature", "pred_cls", "aux_outputs"}
    assert out["aux_outputs"] == []
    for k in ("pred_masks", "pred_cls", "pred_boxes", "pred_logit", "text_sentence_feature"):
        assert tuple(out[k].shape) == g[k].shape, k
    assert maxdiff(out["pred_masks"], g["pred_masks"]) < 1e-3
    assert np.array_equal(out["pred_masks"].numpy() > 0, g["pred_masks"] > 0)
    assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4
    assert maxdiff(out["pred_boxes"], g["pred_boxes"]) < 1e-5
    assert maxdiff(out["pred_logit"], g["pred_logit"]) < 1e-4
    assert maxdiff(out["text_sentence_feature"], g["text_sentence_feature"]) < 1e-4


def run_padded_b2(model, g, device="cpu"):
    T = int(g["T"])
    clips = [W.synthetic_clip(int(seed), T, int(h), int(w)) for seed, (h, w) in zip(g["seeds"], g["sizes"])]
    samples = S.nested_tensor_from_videos_list(clips).to(device)
    assert not samples.unpadded and bool(samples.mask.any())
    ids, attn = torch.from_numpy(g["ids"]).to(device), torch.from_numpy(g["attn"]).to(device)
    targets = [[{"size": torch.tensor([int(h), int(w)])} for h, w in g["sizes"]] for _ in range(T)]
    return model(samples, None, {"input_ids": ids, "attention_mask": attn}, targets)


def check_padded_b2(out, g):
    for k in ("pred_masks", "pred_cls", "pred_boxes", "pred_logit", "text_sentence_feature"):
        assert tuple(out[k].shape) == g[k].shape, k
    assert maxdiff(out["pred_masks"], g["pred_masks"]) < 1e-3
    flips = (out["pred_masks"].cpu().numpy() > 0) != (g["pred_masks"] > 0)
    assert flips.sum() <= 2 and (not flips.any() or np.abs(g["pred_masks"][flips]).max() < 1e-4)
    assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4
    assert maxdiff(out["pred_boxes"], g["pred_boxes"]) < 1e-5
    assert maxdiff(out["pred_logit"], g["pred_logit"]) < 1e-4
    assert maxdiff(out["text_sentence_feature"], g["text_sentence_feature"]) < 1e-4


def test_plumbing_padded_batch_of_two(cpu_model, oracle_kernels, golden):
    """B = 2, clips of different size (frame padding, valid ratios < 1, K2 pad mask) and expressions of
    different length (word padding mask in vlf / lvf / the sentence feature)."""
    g = golden("padded_b2_forward.npz")
    check_padded_b2(run_padded_b2(cpu_model, g), g)


def test_expression_padding_does_not_change_the_outputs(synthetic_sd):
    """<pad> tokens with attention mask 0 behind the expression leave every output unchanged (exactly so for the
    reference, checked in the build container): ClipInferencer relies on it to key hipGraphs on (T, H, W) only."""
    from oracle import soc_oracle as O
    T, H, Wd = 2, 64, 96
    clip, ids = W.synthetic_clip(5, T, H, Wd), W.synthetic_token_ids(5, 6)
    a = O.soc_forward(synthetic_sd, clip, ids, torch.ones_like(ids), (H, Wd))
    padded = torch.cat([ids, torch.ones(1, 4, dtype=torch.long)], 1)
    attn = torch.cat([torch.ones_like(ids), torch.zeros(1, 4, dtype=torch.long)], 1)
    b = O.soc_forward(synthetic_sd, clip, padded, attn, (H, Wd))
    for k in ("pred_masks", "pred_cls", "pred_boxes", "pred_logit", "text_sentence_feature"):
        assert maxdiff(a[k], b[k]) < 1e-6, k


def test_plumbing_t10_temporal_shift(cpu_model, oracle_kernels, golden):
    """T=10 > window: temporal shift 4 and D padded to 16 ('next')."""
    from tests.golden_utils import sub
    g = golden("t10_forward.npz")
    out = run_cfg(cpu_model, g)
    scale = g["pred_masks_stats"][2]
    assert maxdiff(sub(out["pred_masks"], 65536), g["pred_masks_sub"]) < 1e-3, scale
    assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4
    assert maxdiff(out["pred_boxes"], g["pred_boxes"]) < 1e-5


def test_postprocessing_matches_reference_driver(golden):
    from neurips2023_soc_amd import postprocessing as P
    g = golden("full_forward.npz")
    out = {"pred_cls": t(g["pred_cls"]), "pred_masks": None}
    scores = out["pred_cls"][:, 0].sigmoid().mean(0).max(-1)[0]
    assert int(scores.argmax()) == int(g["selected_query"])
    masks = t(g["selected_masks"])
    up = P.upsample_and_threshold(masks, (720, 1280))
    assert up.shape == (8, 720, 1280) and up.dtype == torch.bool
    # sigmoid(x) > 0.5  <=>  x > 0 on the interpolated logits
    ref = torch.nn.functional.interpolate(masks[None], size=(720, 1280), mode="bilinear", align_corners=False)[0] > 0
    assert torch.equal(up, ref)
    lab = P.merge_davis_objects(torch.stack([masks.sigmoid(), (-masks).sigmoid()]))
    assert lab.shape == masks.shape and int(lab.max()) <= 2


def test_nested_tensor_from_videos_list_pads_and_masks():
    a, b = torch.ones(2, 3, 4, 5), torch.ones(3, 3, 2, 6)
    nt = S.nested_tensor_from_videos_list([a, b])
    assert nt.tensors.shape == (3, 2, 3, 4, 6) and nt.mask.shape == (3, 2, 4, 6)
    assert not nt.mask[:2, 0, :4, :5].any() and nt.mask[2, 0].all() and nt.mask[:, 0, :, 5].all()
    assert not nt.mask[:, 1, :2, :].any() and nt.mask[:, 1, 2:, :].all()
    assert float(nt.tensors[2, 0].abs().sum()) == 0


def test_ms_deform_attn_forward_step_check():
    from neurips2023_soc_amd.ms_deform_attn import ms_deform_attn_forward
    v = torch.zeros(3, 4, 8, 32)
    with pytest.raises(RuntimeError, match="must divide"):
        ms_deform_attn_forward(v, None, None, None, None, 2)


def test_c_abi_library_exports_every_declared_symbol():
    """libsoc_hip.so loads without a GPU and exports everything include/soc_hip.h declares."""
    import re
    from neurips2023_soc_amd import _lib, build_ext
    build_ext.build(verbose=False)
    lib = _lib.load()
    hdr = open(os.path.join(os.path.dirname(_lib._PKG), "include", "soc_hip.h")).read()
    declared = set(re.findall(r"\b(soc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.soc_hip_abi_version() == _lib.ABI_VERSION == 16
    # ABI 16 is stateless: no entry point sets or reads process-wide state (ABI <= 15 had soc_*_set_split / soc_set_reserved_cus)
    assert not [n for n in declared if re.search(r"_(set|get)_", n)], declared
    assert lib.soc_mlp_split_max_hidden(512) >= 2048 and lib.soc_mlp_split_max_hidden(256) >= 2048      # the BASELINE layers
    assert lib.soc_mlp_split_max_hidden(100) == 0
    # the routing predicates read a Python table instead of the library (they must work on a CPU-only checkout): same numbers
    from neurips2023_soc_amd import hot_ops
    for c in (64, 96, 100, 128, 192, 256, 384, 512, 768, 1024):
        assert hot_ops.mlp_split_max_hidden(c) == lib.soc_mlp_split_max_hidden(c), c
    assert lib.soc_xattn_workspace_bytes(240, 10, 1, 8, 32) == 0
    assert lib.soc_xattn_workspace_bytes(10, 1920, 1, 8, 32) == 0


def test_msda_fused_rejects_maps_beyond_32_bit_tap_offsets():
    """The fused MSDA entry point forms tap addresses as 32-bit byte offsets inside a frame (include/soc_hip.h):
    value maps with S * M * 128 >= 2^31 or S >= 2^24 must come back SOC_EUNSUPPORTED -- decided from the sizes alone,
    before any launch, so this runs without a GPU (the arguments are never dereferenced)."""
    import ctypes as C
    from neurips2023_soc_amd import _lib, build_ext
    build_ext.build(verbose=False)
    lib = _lib.load()
    buf = (C.c_char * 64)()
    p = C.cast(buf, C.c_void_p)
    M, D, L, P = 8, 32, 4, 4
    for S in (1 << 24, (1 << 31) // (M * 128)):
        code = lib.soc_msda_fused_fwd_f32(p, None, None, p, p, p, 2, p, p, p, 1, S, M, D, L, 5, P, None)
        assert code == _lib.SOC_EUNSUPPORTED, (S, code)


def test_k24_plan_cuts_rows_and_column_spans_legally():
    """soc_xs_linear_plan is a host function (256 CUs without a device): the cut it returns is one soc_xs_linear_f32 accepts --
    ncr column spans of N / 16 / ncr tiles that a built range width (4 / 6 / 8 / 12 / 16 / 18; K <= 256: even; K = 1024: <= 8)
    divides, at most 2048 columns per span (the span's bias waits in LDS), no more workgroup rows than row tiles -- and at the
    row counts of a launch group the rows alone fill the chip, so N is not cut finer than the span limit asks (round 5)."""
    import ctypes as C
    from neurips2023_soc_amd import _lib
    lib = _lib.load()
    built = {192: (4, 6, 8, 12, 16, 18), 256: (4, 6, 8, 12, 16, 18), 384: (4, 6, 8, 12, 16, 18), 512: (4, 6, 8, 12, 16, 18),
             768: (4, 6, 8, 12, 16, 18), 1024: (4, 6, 8)}
    for M in (17, 1920, 7360, 29440, 73600, 115200, 385600):
        for N, K in ((1152, 384), (384, 384), (2304, 768), (768, 768), (3072, 768), (384, 768), (576, 192), (256, 256), (256, 384),
                     (1536, 512), (3072, 1024), (4096, 1024)):
            nrg, ncr, span = C.c_int(0), C.c_int(0), C.c_int(0)
            assert lib.soc_xs_linear_plan(M, N, K, C.byref(nrg), C.byref(ncr), C.byref(span), None) == 0, (M, N, K)
            nrg, ncr, span = nrg.value, ncr.value, span.value
            assert ncr * span == N // 16 and span * 16 <= 2048, (M, N, K, nrg, ncr, span)
            assert any(span % t == 0 and (K > 256 or t % 2 == 0) for t in built[K]), (M, N, K, span)
            assert 1 <= nrg <= (M + 15) // 16, (M, N, K, nrg)
            if M >= 29440 and N <= 2048 and K <= 768:
                assert ncr <= 2, (M, N, K, nrg, ncr, span)          # a group's rows fill 256 CUs: whole-width (or half) spans
    assert lib.soc_xs_linear_plan(100, 40, 384, C.byref(C.c_int()), C.byref(C.c_int()), C.byref(C.c_int()), None) != 0   # N % 32
