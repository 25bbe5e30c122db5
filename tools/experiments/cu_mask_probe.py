"""Do CU-masked streams (hipExtStreamCreateWithCUMask) work on this stack -- eagerly, and under hipGraph replay?

1. which CUs a mask selects (cu_id_probe.hip: HW_ID / XCC_ID per workgroup) for a few masks;
2. a one-workgroup-per-CU persistent kernel (K23, 32 768 x 256 x 2048) on the default stream, on a stream masked to half the
   CUs (2x if the mask is honoured) and on one masked to all but 16;
3. the same through a torch CUDAGraph captured on / replayed into the masked stream.
Also: is a forward bit-identical across host threads / streams (tests/test_gpu_forward.py two-thread test)?
"""
import ctypes as C
import json
import os
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurips2023_soc_amd import hot_ops  # noqa: E402

hip = C.CDLL("libamdhip64.so")
probe = C.CDLL(os.path.join(ROOT, "tools", "experiments", "_build", "libcu_id_probe.so"))
out = {}


def masked_stream(bits):
    """bits: iterable of CU indices set in the mask"""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value), st


def where(stream, blocks=256, lds=150 * 1024):
    buf = torch.zeros(2 * blocks, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = probe.cu_id_launch(C.c_void_p(buf.data_ptr()), blocks, 64, lds, 200000, C.c_void_p(stream.cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    v = buf.cpu().view(blocks, 2)
    ids = set()
    for hw, xcc in v.tolist():
        hw &= 0xFFFFFFFF
        ids.add((xcc & 0xF, (hw >> 13) & 0x7, (hw >> 12) & 1, (hw >> 8) & 0xF))
    return sorted(ids)


dflt = torch.cuda.current_stream()
ids = where(dflt)
out["default_stream_distinct_cus"] = len(ids)
out["default_stream_xccs"] = sorted({i[0] for i in ids})
for name, bits in (("first32", range(32)), ("every8th", range(0, 256, 8)), ("first128", range(128)), ("all_but_16", range(16, 256))):
    s, _h = masked_stream(bits)
    ids = where(s)
    per_xcc = {}
    for x in ids:
        per_xcc[x[0]] = per_xcc.get(x[0], 0) + 1
    out["mask_" + name] = {"distinct_cus": len(ids), "per_xcc": per_xcc}

# 2. K23 timing
g = torch.Generator().manual_seed(0)
M, Cw, F = 32768, 256, 2048
x = torch.randn(M, Cw, generator=g).cuda()
w1, b1 = (torch.randn(F, Cw, generator=g) / 16).cuda(), torch.randn(F, generator=g).cuda()
w2, b2 = (torch.randn(Cw, F, generator=g) / 45).cuda(), torch.randn(Cw, generator=g).cuda()
run = lambda: hot_ops.mlp_split(x, w1, b1, w2, b2, "relu", residual=x)   # noqa: E731
ref = run()
torch.cuda.synchronize()


def time_on(stream, fn, reps=10):
    with torch.cuda.stream(stream):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


half, _h1 = masked_stream(range(128))
most, _h2 = masked_stream(range(16, 256))
out["k23_us"] = {"default": time_on(dflt, run), "mask_128": time_on(half, run), "mask_240": time_on(most, run)}
with torch.cuda.stream(half):
    got = run()
torch.cuda.synchronize()
out["k23_masked_equal"] = bool(torch.equal(got, ref))

# 3. graph
for name, st in (("mask_128", half), ("mask_240", most), ("plain", torch.cuda.Stream())):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        run()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=st):
            y = run()
    torch.cuda.synchronize()
    out.setdefault("k23_graph_us", {})[name + "_replayed_on_itself"] = time_on(st, gr.replay)
    out["k23_graph_us"][name + "_replayed_on_default"] = time_on(dflt, gr.replay)
    out["k23_graph_us"][name + "_replayed_on_mask128"] = time_on(half, gr.replay)

# 4. two graphs in two streams concurrently (round 1: "concurrent graphs hang")
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
s1, s2 = masked_stream(range(16, 256))[0], masked_stream(range(16))[0]
xs = torch.randn(4096, 256, device="cuda")
small = lambda: hot_ops.add_layernorm(xs, xs, torch.ones(256, device="cuda"), torch.zeros(256, device="cuda"), 1e-5)   # noqa: E731
try:
    small()
    with torch.cuda.stream(s1):
        with torch.cuda.graph(g1, stream=s1):
            run()
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g2, stream=s2):
            for _ in range(40):
                small()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        with torch.cuda.stream(s1):
            g1.replay()
        with torch.cuda.stream(s2):
            g2.replay()
    s1.synchronize()
    s2.synchronize()
    b.record()
    torch.cuda.synchronize()
    out["two_graphs_concurrent_us_per_pair"] = a.elapsed_time(b) / 10 * 1e3
    out["g1_alone_us"] = time_on(s1, g1.replay)
    out["g2_alone_us"] = time_on(s2, g2.replay)
    # same pair, both on plain streams
    p1, p2 = torch.cuda.Stream(), torch.cuda.Stream()
    a.record()
    for _ in range(10):
        with torch.cuda.stream(p1):
            g1.replay()
        with torch.cuda.stream(p2):
            g2.replay()
    p1.synchronize()
    p2.synchronize()
    b.record()
    torch.cuda.synchronize()
    out["two_graphs_plain_streams_us_per_pair"] = a.elapsed_time(b) / 10 * 1e3
except Exception as exc:      # noqa: BLE001
    out["two_graphs_error"] = repr(exc)
print(json.dumps(out, indent=1), flush=True)

# 5. forward determinism across threads / streams
import neurips2023_soc_amd as S  # noqa: E402
from neurips2023_soc_amd import weights as W  # noqa: E402
model, _, _ = S.build_model(S.default_args(text_encoder_random_init=True))
W.load_synthetic(model, 2023)
model = model.cuda().eval()
T, H, Wd, L = 8, 360, 640, 10
clip = W.synthetic_clip(1, T, H, Wd)
ids_ = W.synthetic_token_ids(1, L)


def fwd():
    samples = S.nested_tensor_from_videos_list([clip]).to("cuda")
    o = model(samples, None, {"input_ids": ids_, "attention_mask": torch.ones_like(ids_)}, [[{"size": torch.tensor([H, Wd])}] for _ in range(T)])
    torch.cuda.synchronize()
    return {k: o[k].clone() for k in ("pred_masks", "pred_cls", "text_sentence_feature")}


base = fwd()
res = {"main_again": fwd()}
with torch.cuda.stream(torch.cuda.Stream()):
    res["main_new_stream"] = fwd()


def worker(key, new_stream):
    if new_stream:
        with torch.cuda.stream(torch.cuda.Stream()):
            res[key] = fwd()
    else:
        res[key] = fwd()


for key, ns in (("thread_default_stream", False), ("thread_new_stream", True)):
    t = threading.Thread(target=worker, args=(key, ns))
    t.start()
    t.join()
det = {k: {n: float((v[n] - base[n]).abs().max()) for n in base} for k, v in res.items()}
print(json.dumps({"forward_determinism_max_abs_diff_vs_first": det}, indent=1))
