"""-m gpu: the whole HIP hot path (SOC.forward on an MI355X) against the reference's golden
outputs and the CPU oracle.  Tolerance: mask-logit max-abs-diff < 1e-3 (BASELINE.json north_star),
thresholded masks and the selected query bit-exact."""
import numpy as np
import pytest
import torch

import neurips2023_soc_amd as S
from neurips2023_soc_amd import postprocessing as P, weights as W
from tests.golden_utils import sub, t

pytestmark = pytest.mark.gpu

# A thresholded pixel may differ from the reference only where the REFERENCE logit is this close to zero
# (the reference's own 1-vs-8-thread noise is 6e-5 at the BASELINE logit scale): "bit-exact masks" holds
# everywhere except for pixels whose reference logit lies inside the reference's own thread-count noise.
FLIP_WINDOW = 6e-5


def maxdiff(a, b):
    return float((torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


@pytest.fixture(scope="module")
def gpu_model(synthetic_sd):
    assert torch.cuda.is_available()
    model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
    model.load_state_dict(synthetic_sd, strict=False)
    return model.cuda().eval()


def run_cfg(model, cfg):
    seed, T, H, Wd, L = (int(v) for v in cfg)
    samples = S.nested_tensor_from_videos_list([W.synthetic_clip(seed, T, H, Wd)]).to("cuda")
    ids = W.synthetic_token_ids(seed, L)
    targets = [[{"size": torch.tensor([H, Wd])}] for _ in range(T)]
    out = model(samples, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)}, targets)
    torch.cuda.synchronize()
    return out


def test_tiny_config_matches_reference(gpu_model, golden):
    g = golden("tiny_forward.npz")
    out = run_cfg(gpu_model, g["cfg"])
    d = maxdiff(out["pred_masks"], g["pred_masks"])
    print("tiny: max|dlogit|", d, "of", np.abs(g["pred_masks"]).max())
    assert d < 1e-3
    flip = (out["pred_masks"].cpu().numpy() > 0) != (g["pred_masks"] > 0)
    assert flip.sum() <= 2 and (not flip.any() or np.abs(g["pred_masks"][flip]).max() < FLIP_WINDOW)
    assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4
    assert maxdiff(out["pred_boxes"], g["pred_boxes"]) < 1e-5
    assert maxdiff(out["pred_logit"], g["pred_logit"]) < 1e-4
    assert maxdiff(out["text_sentence_feature"], g["text_sentence_feature"]) < 1e-4


def test_full_config_matches_reference(gpu_model, golden):
    """BASELINE config: Swin-T, T=8, 360x640."""
    g = golden("full_forward.npz")
    out = run_cfg(gpu_model, g["cfg"])
    idx, masks = P.select_trajectory(out)
    assert int(idx) == int(g["selected_query"])
    d = maxdiff(masks, g["selected_masks"])
    dsub = maxdiff(sub(out["pred_masks"], 1 << 17), g["pred_masks_sub"])
    print("full: max|dlogit| selected", d, "all(sub)", dsub, "of", g["pred_masks_stats"][2])
    assert d < 1e-3 and dsub < 1e-3
    # thresholded masks: bit-exact except at the decision boundary itself -- a pixel may only
    # differ if the REFERENCE logit is within fp32 noise of zero (|logit| < FLIP_WINDOW on a scale of
    # 37; the reference's own 1-vs-8-thread noise is 1.6e-6 relative = 6e-5 here)
    ours = (out["pred_masks"] > 0).cpu().numpy().reshape(-1)
    ref_bits = np.unpackbits(g["pred_masks_signbits"])[:ours.size].astype(bool)
    flipped = np.nonzero(ours != ref_bits)[0]
    print("full: thresholded-mask flips", flipped.size, "of", ours.size)
    assert flipped.size <= 8
    near = dict(zip(g["near_zero_idx"].tolist(), g["near_zero_val"].tolist()))
    for i in flipped.tolist():
        assert i in near and abs(near[i]) < FLIP_WINDOW, (i, near.get(i))
    assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4
    assert maxdiff(out["pred_boxes"], g["pred_boxes"]) < 1e-5
    assert maxdiff(out["pred_logit"], g["pred_logit"]) < 1e-4


def test_full_config_f32_mfma_mode_matches_reference_and_split_mode(gpu_model, golden):
    """model.matmul_mode = "f32" (default from SOC_MATMUL): every product on the f32-input MFMA, no bf16 matrix-core kernel launched; the
    default split mode launches K20 and the split K1.  Both meet the reference, and differ from each other by f32 noise."""
    from neurips2023_soc_amd import hot_ops
    g = golden("full_forward.npz")
    assert hot_ops.matmul_mode() == "split" and gpu_model.matmul_mode is None
    hot_ops.profile_begin()
    out_split = run_cfg(gpu_model, g["cfg"])
    prof_split = hot_ops.profile_end()
    gpu_model.matmul_mode = "f32"
    try:
        hot_ops.profile_begin()
        out_f32 = run_cfg(gpu_model, g["cfg"])
        prof_f32 = hot_ops.profile_end()
    finally:
        gpu_model.matmul_mode = None
    assert hot_ops.matmul_mode() == "split"                     # the mode lived inside the model's forward only
    assert prof_split.get("linear_split", {}).get("launches", 0) >= 4 and "linear_split" not in prof_f32
    for fam in ("mlp_split", "xs_linear"):                      # K23 / K24 exist in the split arithmetic only
        assert prof_split.get(fam, {}).get("launches", 0) >= 6 and fam not in prof_f32, fam
    for out in (out_split, out_f32):
        idx, masks = P.select_trajectory(out)
        assert int(idx) == int(g["selected_query"])
        assert maxdiff(masks, g["selected_masks"]) < 1e-3
        assert maxdiff(sub(out["pred_masks"], 1 << 17), g["pred_masks_sub"]) < 1e-3
        assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4
    d = maxdiff(out_split["pred_masks"], out_f32["pred_masks"].cpu())
    print("split vs f32 MFMA: max|dlogit|", d)
    assert 0 < d < 5e-4                                         # two different arithmetics, both f32-grade


def test_two_models_with_different_modes_from_two_threads(gpu_model, synthetic_sd, golden, monkeypatch):
    """ABI 16 is stateless: the arithmetic mode is an argument of each launch and a thread-local of each forward.  A "split"
    model and an "f32" model run concurrently from two threads (own streams), three forwards each.  Checked per thread: which
    kernels it launched (the bf16-split-only kernels K20 / K23 / K24 from the "split" thread only; the `split` argument K1 and
    K13 were called with), and that every result meets the reference golden and equals what the same model produces alone to
    within the forward's run-to-run noise (a few library kernels accumulate with atomics: ~4e-5, see the soak test)."""
    import threading
    from neurips2023_soc_amd import _lib, hot_ops
    g = golden("full_forward.npz")
    other, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
    other.load_state_dict(synthetic_sd, strict=False)
    other = other.cuda().eval()
    other.matmul_mode = "f32"
    models = {"split": gpu_model, "f32": other}
    alone = {k: run_cfg(m, g["cfg"])["pred_masks"].clone() for k, m in models.items()}
    d_modes = maxdiff(alone["split"], alone["f32"].cpu())
    assert 0 < d_modes < 5e-4

    seen = {}          # thread ident -> {"split_only": launches of K20 / K23 / K24, "k1": set of split args, "k13": set of split args}
    lib = _lib.load()

    def note(key, value=None):
        rec = seen.setdefault(threading.get_ident(), {"split_only": 0, "k1": set(), "k13": set()})
        if value is None:
            rec[key] += 1
        else:
            rec[key].add(value)

    for name in ("linear_split", "mlp_split", "xs_linear"):
        real = getattr(hot_ops, name)
        monkeypatch.setattr(hot_ops, name, lambda *a, _real=real, **kw: (note("split_only"), _real(*a, **kw))[1])
    real_k1, real_k13 = lib.soc_win_attn3d_f32, lib.soc_ws_linear_f32
    monkeypatch.setattr(lib, "soc_win_attn3d_f32", lambda *a: (note("k1", int(a[-2])), real_k1(*a))[1])
    monkeypatch.setattr(lib, "soc_ws_linear_f32", lambda *a: (note("k13", int(a[-2])), real_k13(*a))[1])

    results, errors, idents = {"split": [], "f32": []}, [], {}
    start = threading.Barrier(2)

    def worker(name):
        try:
            idents[name] = threading.get_ident()
            stream = torch.cuda.Stream()
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(3):
                    results[name].append(run_cfg(models[name], g["cfg"])["pred_masks"].clone())
            stream.synchronize()
        except BaseException as exc:        # noqa: BLE001 -- reported by the main thread
            errors.append((name, exc))

    threads = [threading.Thread(target=worker, args=(n,)) for n in models]
    for t_ in threads:
        t_.start()
    for t_ in threads:
        t_.join()
    torch.cuda.synchronize()
    assert not errors, errors
    s_rec, f_rec = seen[idents["split"]], seen[idents["f32"]]
    assert s_rec["split_only"] >= 3 * 20 and s_rec["k1"] == {1} and s_rec["k13"] == {1}, s_rec
    assert f_rec["split_only"] == 0 and f_rec["k1"] == {0} and f_rec["k13"] <= {0}, f_rec
    for name in models:
        assert len(results[name]) == 3
        for r in results[name]:
            assert maxdiff(r, alone[name].cpu()) < 1e-4, name
            assert maxdiff(sub(r, 1 << 17), g["pred_masks_sub"]) < 1e-3


@pytest.fixture(scope="module")
def gpu_model_b(ref_shapes):
    shapes = {k: v[0] for k, v in ref_shapes("b").items() if v[1].startswith("float")}
    model, _, _ = S.build_model(S.default_args("video-swin-b", text_encoder_random_init=True))
    model.load_state_dict(W.synthetic_state_dict(shapes, seed=2023), strict=False)
    return model.cuda().eval()


@pytest.mark.parametrize("fixture", ["full_forward_b.npz", "full_forward_b720.npz"])
def test_swin_b_configs_match_reference(gpu_model_b, golden, fixture):
    """BASELINE configs 4/5: Video-Swin-B at 360x640 and at 720x1280 (S = 19 160, 180x320 masks)."""
    g = golden(fixture)
    out = run_cfg(gpu_model_b, g["cfg"])
    idx, masks = P.select_trajectory(out)
    assert int(idx) == int(g["selected_query"])
    d = maxdiff(masks, g["selected_masks"])
    dsub = maxdiff(sub(out["pred_masks"], 1 << 17), g["pred_masks_sub"])
    print(fixture, "max|dlogit| selected", d, "all(sub)", dsub, "of", g["pred_masks_stats"][2])
    assert d < 1e-3 and dsub < 1e-3
    flip = (masks.cpu().numpy() > 0) != (g["selected_masks"] > 0)
    assert flip.sum() <= 8 and (not flip.any() or np.abs(g["selected_masks"][flip]).max() < FLIP_WINDOW)
    assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4
    assert maxdiff(out["pred_boxes"], g["pred_boxes"]) < 1e-5


def test_swin_s_matches_reference(golden, ref_shapes):
    """Video-Swin-S backbone (18 blocks in stage 2; reference models/video_swin_transformer.py:749-763) through the
    HIP path against the reference's own forward on a 180x320 clip."""
    shapes = {k: v[0] for k, v in ref_shapes("s").items() if v[1].startswith("float")}
    model, _, _ = S.build_model(S.default_args("video-swin-s", text_encoder_random_init=True))
    model.load_state_dict(W.synthetic_state_dict(shapes, seed=2023), strict=False)
    model = model.cuda().eval()
    g = golden("full_forward_s.npz")
    out = run_cfg(model, g["cfg"])
    idx, masks = P.select_trajectory(out)
    assert int(idx) == int(g["selected_query"])
    assert maxdiff(masks, g["selected_masks"]) < 1e-3
    assert maxdiff(sub(out["pred_masks"], 1 << 14), g["pred_masks_sub"]) < 1e-3
    ours = (out["pred_masks"] > 0).cpu().numpy().reshape(-1)
    ref_bits = np.unpackbits(g["pred_masks_signbits"])[:ours.size].astype(bool)
    near = dict(zip(g["near_zero_idx"].tolist(), g["near_zero_val"].tolist()))
    flipped = np.nonzero(ours != ref_bits)[0]
    assert flipped.size <= 8 and all(i in near and abs(near[i]) < FLIP_WINDOW for i in flipped.tolist())
    assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4
    assert maxdiff(out["pred_boxes"], g["pred_boxes"]) < 1e-5


def test_t10_temporal_shift_matches_reference(gpu_model, golden):
    g = golden("t10_forward.npz")
    out = run_cfg(gpu_model, g["cfg"])
    assert maxdiff(sub(out["pred_masks"], 65536), g["pred_masks_sub"]) < 1e-3
    assert maxdiff(out["pred_cls"], g["pred_cls"]) < 1e-4


def test_forward_repeatable_and_input_mutation_tolerated(gpu_model, golden):
    """Two runs agree to rounding (library GEMM/conv algorithm selection may differ between the
    first and later calls, so not bitwise); bitwise repeatability of the four HIP kernels
    themselves is asserted in test_gpu_kernels.py::test_kernels_bitwise_repeatable."""
    g = golden("tiny_forward.npz")
    a = run_cfg(gpu_model, g["cfg"])["pred_masks"]
    b = run_cfg(gpu_model, g["cfg"])["pred_masks"]
    assert maxdiff(a, b.cpu()) < 5e-4
    assert torch.equal(a > 0, b > 0)


def test_upsample_threshold_on_device(gpu_model, golden):
    g = golden("full_forward.npz")
    m = t(g["selected_masks"]).cuda()
    up = P.upsample_and_threshold(m, (720, 1280))
    ref = P.upsample_and_threshold(m.cpu(), (720, 1280))
    assert float((up.cpu() != ref).float().mean()) < 1e-5


def test_hip_graph_replay_matches_eager(gpu_model, golden):
    """ClipGraph (hipGraph capture of forward + query selection) reproduces the eager forward and
    the packed result record on two different clips replayed through the same graph."""
    from neurips2023_soc_amd import clip_parallel as CP
    from neurips2023_soc_amd.graph_runner import ClipGraph
    T, H, Wd, L = 3, 250, 300, 10
    runner = ClipGraph(gpu_model, T, H, Wd, L, "cuda")
    for seed in (7, 8):
        clip = W.synthetic_clip(seed, T, H, Wd).cuda()
        ids = W.synthetic_token_ids(seed, L).cuda()
        out = runner.run(clip, ids)
        torch.cuda.synchronize()
        rec = runner.record.clone()
        samples = S.nested_tensor_from_videos_list([clip])
        eager = gpu_model(samples, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)},
                          [[{"size": (H, Wd)}]] * T)
        assert maxdiff(out["pred_masks"], eager["pred_masks"].cpu()) < 5e-4
        q, cls, masks = CP.unpack_record(rec.cpu(), T, 20, 63, 75)
        idx, emasks = P.select_trajectory(eager)
        assert q == int(idx)
        assert maxdiff(masks, emasks.cpu()) < 5e-4
    g = golden("tiny_forward.npz")   # seed 7 is the tiny golden clip
    out = runner.run(W.synthetic_clip(7, T, H, Wd).cuda(), W.synthetic_token_ids(7, L).cuda())
    assert maxdiff(out["pred_masks"], g["pred_masks"]) < 1e-3


def test_clip_inferencer_end_to_end(gpu_model, golden):
    """infer.ClipInferencer = reference loop body: graph forward -> best query -> K6 upsample+threshold."""
    from neurips2023_soc_amd.infer import ClipInferencer
    g = golden("full_forward.npz")
    seed, T, H, Wd, L = (int(v) for v in g["cfg"])
    run = ClipInferencer(gpu_model)
    res = run(W.synthetic_clip(seed, T, H, Wd).cuda(), W.synthetic_token_ids(seed, L).cuda(), (720, 1280))
    assert int(res["query"]) == int(g["selected_query"])
    assert maxdiff(res["mask_logits"], g["selected_masks"]) < 1e-3
    want = torch.nn.functional.interpolate(t(g["selected_masks"])[None], size=(720, 1280), mode="bilinear",
                                           align_corners=False)[0]
    diff = res["masks"].cpu() != (want > 0)
    assert res["masks"].shape == (T, 720, 1280)
    assert int(diff.sum()) <= 40 and (not diff.any() or float(want[diff].abs().max()) < 1e-3)


def test_padded_batch_of_two_matches_reference(gpu_model, golden):
    """B = 2 with frame and word padding on the GPU (fused K2 with the pad mask, K3 key padding masks)."""
    from tests.test_host_plumbing import check_padded_b2, run_padded_b2
    g = golden("padded_b2_forward.npz")
    check_padded_b2(run_padded_b2(gpu_model, g, "cuda"), g)


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_odd_geometries_match_reference(gpu_model, golden, case):
    """Clip shapes the drivers can meet but the headline goldens do not cover: a single frame, odd frame counts
    (temporal window clamped / padded), sizes that need padding at every stage, tiny last levels (1x1, 2x2)."""
    g = golden("odd_geometries.npz")
    T, H, Wd, L = (int(v) for v in g["cfgs"][case])
    out = run_cfg(gpu_model, (100 + T, T, H, Wd, L))
    want = g[f"c{case}_pred_masks"]
    assert tuple(out["pred_masks"].shape) == want.shape
    d = maxdiff(out["pred_masks"], want)
    print(f"T={T} {H}x{Wd}: max|dlogit| {d:.2e} of {np.abs(want).max():.1f}")
    assert d < 1e-3
    assert maxdiff(out["pred_cls"], g[f"c{case}_pred_cls"]) < 1e-4
    assert maxdiff(out["pred_boxes"], g[f"c{case}_pred_boxes"]) < 1e-5
    assert maxdiff(out["pred_logit"], g[f"c{case}_pred_logit"]) < 1e-4


def test_clip_inferencer_pads_expressions_for_graph_reuse(gpu_model, golden):
    """Two expressions of different length share one graph (padded to 16 tokens) and give the eager results."""
    from neurips2023_soc_amd.infer import ClipInferencer
    T, H, Wd = 3, 96, 128
    clip = W.synthetic_clip(9, T, H, Wd).cuda()
    eager = ClipInferencer(gpu_model, "cuda", use_graphs=False)
    graphs = ClipInferencer(gpu_model, "cuda", use_graphs=True, pad_tokens_to=16)
    for L in (5, 11):
        ids = W.synthetic_token_ids(30 + L, L).cuda()
        a, b = eager(clip, ids, (H, Wd)), graphs(clip, ids, (H, Wd))
        assert int(a["query"]) == int(b["query"])
        assert maxdiff(b["mask_logits"], a["mask_logits"].cpu()) < 1e-4
        assert float((a["masks"] != b["masks"]).float().mean()) < 1e-4
    assert len(graphs._graphs) == 1


@pytest.mark.parametrize("pipeline", ["two-stream", "one-graph"])
def test_pipelined_graph_matches_plain_graph(gpu_model, pipeline):
    """The software pipelines (tail of clip i beside the head of clip i+1: the one-graph PipelinedClipGraph, what ships, and
    TwoStreamClipGraph) return ClipGraph's records, one call late."""
    from neurips2023_soc_amd.graph_runner import ClipGraph, pipeline_class
    PipelinedClipGraph = pipeline_class(pipeline)
    T, H, Wd, L = 3, 96, 128, 6
    clips = [W.synthetic_clip(40 + i, T, H, Wd).cuda() for i in range(5)]
    ids = [W.synthetic_token_ids(40 + i, L).cuda() for i in range(5)]
    plain = ClipGraph(gpu_model, T, H, Wd, L, "cuda")
    want = []
    for c, t in zip(clips, ids):
        plain.run(c, t)
        want.append(plain.record.clone())
    pipe = PipelinedClipGraph(gpu_model, T, H, Wd, L, "cuda")
    assert pipe.flush() == []

    def through(n):
        got = []
        for c, t in zip(clips[:n], ids[:n]):
            r = pipe.run(c, t)
            if r is not None:
                got.append(r.clone())
        return got + pipe.flush()

    for n in (5, 1, 2, 5):          # incl. streams shorter than the pipeline, and reuse of the state buffers
        got = through(n)
        assert len(got) == n
        for a, b in zip(got, want):
            assert int(a[0]) == int(b[0])                       # selected query
            assert maxdiff(a, b.cpu()) < 1e-4


@pytest.mark.parametrize("pipeline", ["two-stream", "one-graph"])
def test_pipelined_graph_full_config_matches_reference(gpu_model, golden, pipeline):
    """What bench.py times -- the software pipeline at the BASELINE size (T=8, 360x640) -- against the reference's
    own output (full_forward.npz): selected query, its mask logits, class scores; with other clips before and
    after it in the pipeline, and again after a flush (state-buffer reuse)."""
    from neurips2023_soc_amd import clip_parallel as CP
    from neurips2023_soc_amd.graph_runner import pipeline_class
    PipelinedClipGraph = pipeline_class(pipeline)
    g = golden("full_forward.npz")
    seed, T, H, Wd, L = (int(v) for v in g["cfg"])
    hm, wm = -(-H // 4), -(-Wd // 4)
    pipe = PipelinedClipGraph(gpu_model, T, H, Wd, L, "cuda")
    clips = {s_: W.synthetic_clip(s_, T, H, Wd).cuda() for s_ in (seed, seed + 1, seed + 2)}
    ids = W.synthetic_token_ids(seed, L).cuda()
    want = t(g["selected_masks"]).reshape(T, hm, wm)

    def check(rec):
        q, cls, masks = CP.unpack_record(rec.cpu(), T, 20, hm, wm)
        assert q == int(g["selected_query"])
        assert maxdiff(masks, want) < 1e-3
        assert maxdiff(cls, g["pred_cls"].reshape(T, 20)) < 1e-4
        flip = (masks > 0) != (want > 0)
        assert int(flip.sum()) <= 8 and (not bool(flip.any()) or float(want[flip].abs().max()) < FLIP_WINDOW)

    for order in ((seed + 1, seed, seed + 2), (seed, seed + 1), (seed + 2, seed + 1, seed)):
        recs = []
        for s_ in order:
            r = pipe.run(clips[s_], ids)
            if r is not None:
                recs.append(r.clone())
        recs += pipe.flush()
        assert len(recs) == len(order)
        check(recs[order.index(seed)])
        other = recs[order.index(seed + 1)]
        assert maxdiff(other, recs[order.index(seed)].cpu()) > 1e-2       # a different clip gave a different record


@pytest.mark.parametrize("group", [4, 2, 8, 10])
def test_group_pipeline_gives_every_clip_its_single_clip_result(gpu_model, golden, group):
    """The group pipelines (graph_runner.group_pipeline_class: two / four / eight / TEN clips per launch group -- ten is what bench.py
    times), the VOC module per clip.  Every clip's record equals the one-clip ClipGraph's to f32 rounding -- whichever slot it sits
    in, whoever its partners are, with a part-filled last group -- and the golden clip meets the reference's output.  Every slot of a
    group holds a DIFFERENT clip (group + 1 distinct clips, two expressions), so a record that came back from the wrong slot fails.
    (The reference's own B = 2 forward does NOT give a clip its B = 1 result: its VOC couples the batch.)"""
    from neurips2023_soc_amd import clip_parallel as CP
    from neurips2023_soc_amd.graph_runner import ClipGraph, group_pipeline_class
    g = golden("full_forward.npz")
    seed, T, H, Wd, L = (int(v) for v in g["cfg"])
    hm, wm = -(-H // 4), -(-Wd // 4)
    n = max(5, group + 1)
    clips = [W.synthetic_clip(seed + i, T, H, Wd).cuda() for i in range(n)]
    ids = [W.synthetic_token_ids(seed + (i % 2), L).cuda() for i in range(n)]          # two different expressions
    plain = ClipGraph(gpu_model, T, H, Wd, L, "cuda")
    want = []
    for c, t_ in zip(clips, ids):
        plain.run(c, t_)
        want.append(plain.record.clone())
    for i in range(1, n):                                                   # the clips really are distinct
        assert maxdiff(want[i], want[0].cpu()) > 1e-2
    pipe = group_pipeline_class(group)(gpu_model, T, H, Wd, L, "cuda")
    assert pipe.CLIPS == group and pipe.flush() == []

    def through(order):
        """clips in `order` through the group pipeline -> their records, in order"""
        got, counts = [], []
        for j in range(0, len(order), group):
            pair = order[j:j + group]
            for slot, i in enumerate(pair):
                pipe.stage_inputs(clips[i], ids[i], slot=slot)
            counts.append(len(pair))
            r = pipe.replay()
            if r is not None:
                got += [x.clone() for x in r[:counts.pop(0)]]
        for r in pipe.flush():
            got += [x.clone() for x in r[:counts.pop(0)]]
        return got

    everyone = list(range(n))
    for order in (everyone, [1, 0], [4], [3, 0, 2], everyone[::-1] + everyone[1:], everyone[3:] + everyone[:3] + everyone[5:]):
        got = through(order)
        assert len(got) == len(order)
        for i, rec in zip(order, got):
            assert int(rec[0]) == int(want[i][0]), (order, i)
            assert maxdiff(rec, want[i].cpu()) < 1e-4, (order, i)
    q, cls, masks = CP.unpack_record(through([0, 1])[0].cpu(), T, 20, hm, wm)       # clip 0 = the golden's clip and expression
    ref = t(g["selected_masks"]).reshape(T, hm, wm)
    assert q == int(g["selected_query"]) and maxdiff(masks, ref) < 1e-3
    flip = (masks > 0) != (ref > 0)
    assert int(flip.sum()) <= 8 and (not bool(flip.any()) or float(ref[flip].abs().max()) < FLIP_WINDOW)


def test_reference_couples_a_batch_in_voc_only(gpu_model):
    """A B = 2 head + per-clip tails (SOC.split_state) and a fully batched forward with ONLY the VOC module per clip
    (forward_tail(voc_per_clip=True)) both give each clip its single-clip outputs; the plain B = 2 forward -- the reference's
    batch semantics, pinned by padded_b2_forward.npz -- does not."""
    T, H, Wd, L = 3, 96, 128, 6
    clips = [W.synthetic_clip(70 + i, T, H, Wd).cuda() for i in range(2)]
    ids = torch.cat([W.synthetic_token_ids(70 + i, L) for i in range(2)], 0).cuda()
    text = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    targets1 = [[{"size": (H, Wd)}] for _ in range(T)]
    pad = torch.zeros(T, 2, H, Wd, dtype=torch.bool, device="cuda")
    singles = [gpu_model(S.NestedTensor(clips[b][:, None].contiguous(), pad[:, :1], unpadded=True), None,
                         {k: v[b:b + 1] for k, v in text.items()}, targets1) for b in range(2)]
    sb = gpu_model.forward_head(S.NestedTensor(torch.stack(clips, 1), pad, unpadded=True), None, text)
    states = gpu_model.split_state(sb)
    assert len(states) == 2 and all(st["B"] == 1 for st in states)
    for b, st in enumerate(states):
        out = gpu_model.forward_tail(st, targets1)
        scale = float(singles[b]["pred_masks"].abs().max())
        assert maxdiff(out["pred_masks"], singles[b]["pred_masks"].cpu()) < 2e-5 * max(scale, 1.0) + 1e-4
        assert maxdiff(out["pred_cls"], singles[b]["pred_cls"].cpu()) < 1e-4
    targets2 = [[{"size": (H, Wd)}] * 2 for _ in range(T)]
    batched = gpu_model.forward_tail(sb, targets2, voc_per_clip=True)
    for b in range(2):
        scale = float(singles[b]["pred_masks"].abs().max())
        assert maxdiff(batched["pred_masks"][:, b:b + 1], singles[b]["pred_masks"].cpu()) < 2e-5 * max(scale, 1.0) + 1e-4
        assert maxdiff(batched["pred_cls"][:, b:b + 1], singles[b]["pred_cls"].cpu()) < 1e-4
    both = gpu_model(S.NestedTensor(torch.stack(clips, 1), pad, unpadded=True), None, text, targets2)
    assert maxdiff(both["pred_masks"][:, :1], singles[0]["pred_masks"].cpu()) > 1e-3        # the reference's VOC couples the batch


def test_pipelined_graph_soak_full_config():
    """>= 1 500 back-to-back replays of the shipped two-stage pipeline at the BASELINE size: no hang (the child is
    killed and the test fails after the timeout) and no drift: every time a clip comes round its record equals the
    first one to within the run-to-run noise of the library GEMM / conv kernels (~4e-5 on logits of magnitude 37,
    some of them accumulate with atomics), 25x below the parity tolerance, and the noise does not grow with the
    replay count.  Runs in a child process so that a hung GPU queue cannot wedge the test session."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import neurips2023_soc_amd as S
from neurips2023_soc_amd import weights as W
from neurips2023_soc_amd.graph_runner import pipeline_class
PipelinedClipGraph = pipeline_class()          # the shipped one
T, H, Wd, L, N = 8, 360, 640, 10, 1536
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clips = [W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(3)]
ids = W.synthetic_token_ids(1, L).cuda()
pg = PipelinedClipGraph(model, T, H, Wd, L, "cuda")
first, worst, worst_late = {}, torch.zeros((), device="cuda"), torch.zeros((), device="cuda")
for r in range(N):
    rec = pg.run(clips[r %% 3], ids)
    if rec is not None:
        k = (r - 1) %% 3
        if k not in first:
            first[k] = rec.clone()
        else:                                # compared on the device: no host sync in the loop
            d = (rec - first[k]).abs().max()
            worst = torch.maximum(worst, d)
            if r >= N // 2:
                worst_late = torch.maximum(worst_late, d)
last = pg.flush()
torch.cuda.synchronize()
worst, worst_late = float(worst), float(worst_late)
assert len(last) == 1 and len(first) == 3
assert bool(torch.isfinite(last[0]).all())
print("SOAK_OK replays", N, "max deviation from the first record", worst, "in the second half", worst_late)
assert worst < 2e-4 and worst_late <= worst, (worst, worst_late)
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "SOAK_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.parametrize("group,replays", [(4, 400), (10, 160)])
def test_group_pipeline_soak_full_config(group, replays):
    """400 / 160 back-to-back replays (1 600 clips) of the four- / ten-clip launch-group pipelines that bench.py times, at the BASELINE size:
    no hang (child process, killed after the timeout) and no drift -- every time a clip comes round, in whichever slot, its record
    equals the first one to within the run-to-run noise of the library kernels, and the noise does not grow."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import neurips2023_soc_amd as S
from neurips2023_soc_amd import weights as W
from neurips2023_soc_amd.graph_runner import group_pipeline_class
T, H, Wd, L, N, G = 8, 360, 640, 10, %d, %d
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
clips = [W.synthetic_clip(1 + i, T, H, Wd).cuda() for i in range(G + 1)]       # G + 1 clips over G slots: every clip visits every slot
ids = [W.synthetic_token_ids(1 + i %% 2, L).cuda() for i in range(G + 1)]
pg = group_pipeline_class(G)(model, T, H, Wd, L, "cuda")
first, worst, worst_late = {}, torch.zeros((), device="cuda"), torch.zeros((), device="cuda")
order = []
for r in range(N):
    group = [(G * r + b) %% (G + 1) for b in range(G)]
    for b, c in enumerate(group):
        pg.stage_inputs(clips[c], ids[c], slot=b)
    rec = pg.replay()
    order.append(group)
    if rec is not None:
        for b, c in enumerate(order[-2]):
            if c not in first:
                first[c] = rec[b].clone()
            else:
                d = (rec[b] - first[c]).abs().max()
                worst = torch.maximum(worst, d)
                if r >= N // 2:
                    worst_late = torch.maximum(worst_late, d)
last = pg.flush()
torch.cuda.synchronize()
worst, worst_late = float(worst), float(worst_late)
assert len(last) == 1 and last[0].shape[0] == G and len(first) == G + 1
assert bool(torch.isfinite(last[0]).all())
print("SOAK_OK clips per group", G, "replays", N, "max deviation from the first record", worst, "in the second half", worst_late)
assert worst < 2e-4 and worst_late <= worst, (worst, worst_late)
""" % (root, replays, group)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "SOAK_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
