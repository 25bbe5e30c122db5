"""bench.py with another K7 row bound (fused.is_small): python tools/experiments/small_rows_ab.py <rows> [bench args]"""
import runpy
import sys
sys.path.insert(0, ".")
import neurips2023_soc_amd.hot_ops as h  # noqa: E402
h.SMALL_LINEAR_MAX_ROWS = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
