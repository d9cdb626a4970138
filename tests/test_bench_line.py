"""The bench contract, checked on the lines committed under profiles/r03 (they are what `python bench.py` printed on the GPU box):
metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload,
the `roofline` and `cpu_baseline` objects, and this round's additions (executed rays as the headline, repetitions with their spread)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = ["bench.json", "bench_driver_config.json", "bench_dragon871k.json", "bench_cfg2.json", "bench_4k.json"]


@pytest.mark.parametrize("name", LINES)
def test_committed_bench_lines_keep_the_contract(name):
    path = os.path.join(ROOT, "profiles", "r03", name)
    d = json.loads(open(path).read().strip().split("\n")[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and "workload" in d["config"] and "model" not in d["config"]
    # the headline is the work the timed mode executed; the reference-defined rate stands beside it and is larger
    assert d["value"] == d["mrays_executed_per_s"] and d["mrays_reference_defined_per_s"] > d["value"]
    assert abs(d["value"] - d["rays_executed_per_step"] / d["ms_per_step"] / 1e3) < 0.02 * d["value"]
    lo, hi = d["ms_per_step_spread"]
    assert d["repeats"] >= 5 and lo <= d["ms_per_step"] <= hi and hi < 1.1 * lo
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] > 0 and 0 < r["l2"]["hit_rate"] < 1 and 0 < r["hbm"]["traffic_frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and "sample" in c
