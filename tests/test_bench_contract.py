"""CPU-side checks of bench.py: the cgroup quota reader behind cpu_baseline, and the shape of the JSON line the driver
parses -- checked on the line committed under profiles/ (produced on an MI355X by the same bench.py)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("soc_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_granted_cpus_reads_cgroup_v2_and_v1_quotas(tmp_path):
    bench = _bench()
    avail = len(os.sched_getaffinity(0))
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("150000 100000\n")                 # 1.5 CPUs -> 2
    assert bench.granted_cpus(str(v2)) == (min(avail, 2), 2, avail)
    (v2 / "cpu.max").write_text("max 100000\n")
    assert bench.granted_cpus(str(v2)) == (avail, None, avail)
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("1600000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.granted_cpus(str(v1)) == (min(avail, 16), 16, avail)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")           # unlimited
    assert bench.granted_cpus(str(v1)) == (avail, None, avail)
    assert bench.granted_cpus(str(tmp_path / "missing")) == (avail, None, avail)


def test_committed_bench_line_keeps_the_driver_contract():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    with open(os.path.join(ROOT, "profiles", "bench_r04_n1.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "clips/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["n_gpus"] == 1 and line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - line["n_gpus"]) < 1e-6      # clips/s x s/clip = ranks
    assert str(base.get("metric", "")).split()[0].lower() in line["metric"].lower() or "clips" in line["metric"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the roofline object is the kernel family with the largest share of a clip, the others follow in the same form; every one
    # is priced against the ceiling that binds it (bf16 peak / 6 for the split kernels, HBM for the byte-bound ones): a
    # fraction above 1 would mean the wrong ceiling
    others = [line[k] for k in line if k.startswith("roofline_") and k != "roofline_other"]
    assert others and all(r["ms_per_clip"] >= o["ms_per_clip"] for o in others)
    for o in others + [r]:
        assert abs(o["frac"] - o["achieved"] / o["peak"]) < 1e-9 and 0 < o["frac"] <= 1.0, o["kernel"]
        assert 0 < o["frac_of_ceiling"] <= 1.0 and all(0 < sh["frac_of_ceiling"] <= 1.0 for sh in o["per_shape"]), o["kernel"]
        assert o["peak"] in (2500.0 / 6.0, 157.3, 8000.0), (o["kernel"], o["peak"])
        if o["bound"] == "mfma":
            assert o["unit"] == "TFLOP/s" and o["mfma_ms_at_peak"] >= o["hbm_ms_at_6_3_TBs"]
        else:
            assert o["unit"] == "GB/s" and o["peak"] == 8000.0
    assert "mlp_split" in r["kernel"] and any("win_attn3d" in o["kernel"] for o in others)
    assert len(line["parity"]["timed_path_other_records_vs_cpu_oracle"]) == 3
    assert all(e["mask_logit_max_abs_diff"] < 1e-3 and e["selected_query"] == e["selected_query_oracle"]
               for e in line["parity"]["timed_path_other_records_vs_cpu_oracle"])
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["unit"] == "clips/s" and c["cores"] >= 1 and c["sample"]
    assert c["value"] > 0 and line["value"] / c["value"] > 100
    assert line["parity"]["timed_path_mask_logit_max_abs_diff"] < 1e-3
    assert line["parity"]["timed_path_thresholded_mask_flips"] == 0
    assert line["stream_ms_per_step"] > 0 and line["f32_mfma_only_ms_per_step"] > line["ms_per_step"]
