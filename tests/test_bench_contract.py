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


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")


def _check_contract(line):
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line["unit"] == "clips/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - line["n_gpus"]) < 1e-6      # clips/s x s/clip = ranks


def _check_compact(line, text):
    """What the driver must be able to ingest: ONE bounded line (round 4's 21.5 KB line left BENCH_r04.parsed null)."""
    bench = _bench()
    assert len(text) < bench.MAX_LINE_BYTES <= 8000, len(text)
    assert "\n" not in text
    _check_contract(line)
    r = line["roofline"]
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "ms_per_clip", "source"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1.0
    assert r["peak"] in (2500.0 / 6.0, 157.3, 8000.0)
    assert r["ms_per_clip"] < line["ms_per_step"]
    # the roofline object is the kernel family with the largest share of a clip; the others follow as one short row each
    fams = line["roofline_families"]
    assert fams and all(r["ms_per_clip"] >= f["ms_per_clip"] for f in fams.values())
    assert all(0 < f["frac"] <= 1.0 for f in fams.values())
    c = line["cpu_baseline"]
    assert set(c) == {"value", "unit", "cores", "kind", "cpu_model", "sample"}
    assert c["kind"] in ("reference", "port") and c["unit"] == "clips/s" and c["cores"] >= 1 and c["sample"]
    assert c["value"] > 0 and line["value"] / c["value"] > 100
    p = line["parity"]
    # rounds 4-5: four records (golden + three oracle forwards); round 6: EVERY timed record (distinct clips per group slot)
    assert p["records"] >= 4 and p["selected_query_matches"] is True
    if "all_records_max_abs_diff_vs_single_clip" in p:
        assert p["records"] == line["steps"] and p["distinct_clips"] >= min(line["steps"], line["clips_per_head_launch"])
        assert p["all_records_selected_query_equal"] is True and p["all_records_max_abs_diff_vs_single_clip"] < 1e-4
    assert p["mask_logit_max_abs_diff"] < 1e-3
    assert p["flip_window"] == bench.FLIP_WINDOW == 6e-5
    # "bit-exact masks": a thresholded pixel may differ only inside the reference's own thread-count noise of zero
    assert p["flips_total"] == 0 or p["max_abs_ref_logit_at_flips"] < p["flip_window"]


def test_compaction_of_a_long_form_line_fits_the_driver():
    """bench.compact_line on the long form an MI355X run produced (round 4's, 21.5 KB): every contract key survives, the
    result is bounded, and nothing but the long tables is lost."""
    bench = _bench()
    with open(os.path.join(ROOT, "profiles", "bench_r04_n1.json")) as f:
        full = json.loads(f.read().strip().splitlines()[-1])
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(json.dumps(full)) > 20000 and len(text) < 5000
    _check_compact(line, text)
    assert "mlp_split" in line["roofline"]["kernel"] and "win_attn3d" in line["roofline_families"]
    assert line["parity"]["flips_total"] == 2 and line["parity"]["max_abs_ref_logit_at_flips"] < 1e-5


def test_stub_line_is_the_same_bounded_form():
    """`bench.py --stub` prints its line through the same emitter: last line of stdout, parseable, bounded."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, check=True).stdout
    text = out.strip().splitlines()[-1]
    line = json.loads(text)
    assert len(text) < 1000 and line["stub"] is True and line["records_ok"] is True
    _check_contract(line)


import pytest


@pytest.mark.parametrize("name", ["bench_r05_n1.json", "bench_r06_n1.json"])
def test_committed_bench_line_keeps_the_driver_contract(name):
    """The line a round's bench.py printed on an MI355X (committed as it came off stdout)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip(f"{name}: no line committed yet")
    with open(path) as f:
        text = f.read().strip().splitlines()[-1]
    line = json.loads(text)
    _check_compact(line, text)
    assert line["n_gpus"] == 1
    assert line["stream_ms_per_step"] > 0 and line["f32_mfma_only_ms_per_step"] > line["ms_per_step"]
    assert line["detail"].endswith(".json")
    if name >= "bench_r06":
        # round 6: every timed record checked, the roofline says what its peak is, the group does not hang on --steps
        assert line["parity"]["records"] == line["steps"] and "peak_basis" in line["roofline"] and "vs_f32_mfma_peak" in line["roofline"]
        assert line["clips_per_head_launch"] == 10


def test_emit_drops_optional_blocks_instead_of_failing(capsys):
    """A line that grew past the driver's bound loses optional blocks (they are in the detail file) and says so; it never costs
    a finished run its JSON line."""
    bench = _bench()
    line = {"metric": "m", "value": 1.0, "config": {"workload": "w" * 300}, "roofline_families": {str(i): "x" * 100 for i in range(90)},
            "switches": ["s" * 50] * 10, "kernels_per_forward": {"a": 1}}
    bench.emit(dict(line), None, None)
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert "roofline_families" not in out and "roofline_families" in out["truncated"] and out["value"] == 1.0
    assert len(json.dumps(out)) < bench.MAX_LINE_BYTES


def test_default_pipeline_is_a_fixed_group():
    """bench.default_pipeline: DEFAULT_GROUP = 10 clips per launch group whatever --steps is (round 5 picked the first of 8, 10, 12,
    ... that divided --steps -- a group tuned to the driver's 20); fewer clips than a group: one group of them; above 360x640 pairs."""
    import bench
    assert bench.DEFAULT_GROUP == 10
    assert bench.default_pipeline(20, 8, 360, 640) == "group10" and bench.default_pipeline(200, 8, 360, 640) == "group10"
    assert bench.default_pipeline(23, 8, 360, 640) == "group10" and bench.default_pipeline(12, 8, 360, 640) == "group10"
    assert bench.default_pipeline(8, 8, 360, 640) == "octs" and bench.default_pipeline(3, 8, 360, 640) == "group3"
    assert bench.default_pipeline(1, 8, 360, 640) == "one-graph"
    assert bench.default_pipeline(20, 8, 720, 1280) == "pairs"
    from neurips2023_soc_amd import graph_runner
    for name, clips in (("group10", 10), ("octs", 8), ("quads", 4), ("pairs", 2), ("group3", 3), ("one-graph", 1)):
        assert graph_runner.pipeline_class(name).CLIPS == clips
