"""The bench line's `roofline` block is assembled by pure functions (gpuart_amd/bench_line.py): this file drives them on stubbed
counters — fabricated rocprofv3 output directories included — so that a definition that drifts fails here, on the CPU, not in front
of the driver. The lines committed under profiles/ (what `python bench.py` printed on the GPU box) are then checked against the same
contract: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload, the `roofline` and `cpu_baseline` objects."""
import csv
import glob
import json
import os

import pytest

from gpuart_amd import bench_line as BL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_counters(d, rows):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, ["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for k, c, v in rows:
            w.writerow({"Kernel_Name": k, "Counter_Name": c, "Counter_Value": v})


def _write_trace(d, rows):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "x_kernel_trace.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, ["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        for k, a, b in rows:
            w.writerow({"Kernel_Name": k, "Start_Timestamp": a, "End_Timestamp": b})


KT = "void (anonymous namespace)::k_trace<false, 6>(gd::Scene, gd::Frame, gpuart_params)"
KS = "void (anonymous namespace)::k_shade<false>(gd::Scene, gd::Frame)"
KG = "(anonymous namespace)::k_gen(gd::Frame, gpuart_params)"


def test_kernel_names_are_grouped_by_family():
    assert BL.kernel_family(KT) == "k_trace" and BL.kernel_family(KS) == "k_shade" and BL.kernel_family(KG) == "k_gen"
    assert BL.kernel_family("k_accumulate(HIP_vector_type<float, 4u>*, unsigned long)") == "k_accumulate"


def test_counter_sets_are_filtered_by_what_the_box_offers():
    avail = "gfx950 counters: SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY_EXTRA TCC_HIT_sum FETCH_SIZE"
    assert BL.pick_available(["SQ_WAVES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"], avail) == ["SQ_WAVES", "SQ_WAVE_CYCLES"]  # whole words only
    assert BL.pick_available(["TCC_HIT_sum", "TCC_MISS_sum"], avail) == ["TCC_HIT_sum"]
    assert BL.pick_available(["A", "B"], "") == ["A", "B"]  # no listing: ask for the set as it stands
    names = [c for _, cs in BL.PMC_SETS for c in cs]
    assert len(names) == len(set(names))
    assert not ({"FETCH_SIZE", "WRITE_SIZE"} <= set(next(cs for t, cs in BL.PMC_SETS if t == "fetch")))  # never in one pass (TCC slots)
    for must in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        assert must in names


def test_roofline_block_from_stubbed_counters(tmp_path):
    passes = 60  # 3 sequences of 20 passes: what the profiled children render
    # per pass: k_trace 5e8 VALU instructions at 60 % lane utilisation, k_shade 5e7 at 90 %, k_gen 1e7 at 100 %
    rows = []
    for k, valu, lanes in ((KT, 5e8, 0.6), (KS, 5e7, 0.9), (KG, 1e7, 1.0)):
        for _ in range(3):  # three dispatches of every kernel: sums, not lasts
            rows += [(k, "SQ_INSTS_VALU", valu * passes / 3), (k, "SQ_ACTIVE_INST_VALU", valu * passes / 3 * 4),
                     (k, "SQ_THREAD_CYCLES_VALU", valu * passes / 3 * 4 * 64 * lanes), (k, "SQ_INSTS_SALU", valu * passes / 3 * 0.4)]
    _write_counters(str(tmp_path / "valu"), rows)
    _write_counters(str(tmp_path / "wave"), [(KT, "SQ_WAVE_CYCLES", 1000 * passes), (KT, "SQ_ACTIVE_INST_ANY", 400 * passes), (KT, "SQ_WAIT_ANY", 450 * passes),
                                             (KT, "SQ_WAIT_INST_ANY", 150 * passes), (KS, "SQ_WAVE_CYCLES", 100 * passes), (KS, "SQ_ACTIVE_INST_ANY", 50 * passes)])
    _write_counters(str(tmp_path / "tcp"), [(KT, "TCP_TOTAL_CACHE_ACCESSES_sum", 3e8 * passes), (KT, "TCP_TOTAL_ACCESSES_sum", 8e8 * passes), (KT, "TCP_TCC_READ_REQ_sum", 3e7 * passes)])
    _write_counters(str(tmp_path / "tcc"), [(KT, "TCC_HIT_sum", 7e6 * passes), (KT, "TCC_MISS_sum", 3e6 * passes), (KT, "TCC_REQ_sum", 1e7 * passes)])
    _write_counters(str(tmp_path / "fetch"), [(KT, "FETCH_SIZE", 100000 * passes), (KS, "FETCH_SIZE", 300000 * passes)])   # KB
    _write_counters(str(tmp_path / "write"), [(KT, "WRITE_SIZE", 50000 * passes), (KS, "WRITE_SIZE", 250000 * passes)])
    _write_trace(str(tmp_path / "trace"), [(KT, 0, 2_000_000 * passes), (KT, 0, 1_000_000 * passes), (KS, 0, 500_000 * passes), (KG, 0, 100_000 * passes)])
    prof = {t: BL.read_counters(str(tmp_path / t)) for t in ("valu", "wave", "tcp", "tcc", "fetch", "write")}
    trace = BL.read_kernel_trace(str(tmp_path / "trace"))
    assert trace["k_trace"] == (2, 3.0 * passes) and trace["k_gen"][0] == 1
    ms = 0.8
    roof = BL.assemble_roofline(ms, passes, prof, trace, executed={"nodes": 1.6e8, "steps": 0.79e8, "algorithmic_bytes": 8.0e9},
                                reference={"nodes": 2.8e8, "algorithmic_bytes": 15.0e9}, kernel_events=(400.0, 200, 0.1), passes_timed=100)
    # the contract's keys, and the contract's arithmetic: 200 launches in 100 timed passes = 2 per pass, 8e9 algorithmic bytes per pass = 4e9 per
    # launch, 2.0 ms per launch on average -> 2000 GB/s of 8000; traffic = the dominant kernel's own PMC bytes / its 2 profiled launches
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["per_launch"]["launches_per_pass"] == pytest.approx(2.0) and roof["per_launch"]["algorithmic_bytes"] == pytest.approx(4.0e9)
    assert roof["per_launch"]["avg_ms"] == pytest.approx(2.0)
    assert roof["achieved"] == pytest.approx(2000.0) and roof["frac"] == pytest.approx(0.25)
    assert roof["traffic"] == pytest.approx((2 * 100000 + 50000) * 1024.0 * passes / 2)
    # issue slots: (5e8 + 5e7 + 1e7) instructions per pass / 0.8 ms = 700 G/s of 1228.8
    vi = roof["valu_issue"]
    assert vi["instr_per_pass"] == pytest.approx(5.6e8) and vi["achieved"] == pytest.approx(700.0) and vi["frac"] == pytest.approx(700.0 / 1228.8, abs=1e-4)
    # lane utilisation is instruction-weighted over ALL kernels: (5e8*.6 + 5e7*.9 + 1e7) / 5.6e8
    lu = (5e8 * 0.6 + 5e7 * 0.9 + 1e7) / 5.6e8
    assert vi["lane_util"] == pytest.approx(lu, abs=1e-4)
    # useful lane slots (the top-level figure of rounds 3-4, now a block of its own) = issue share x lane utilisation, and achieved / peak says the same
    ls = roof["lane_slots"]
    assert ls["frac"] == pytest.approx(vi["frac"] * lu, abs=2e-4)
    assert ls["frac"] == pytest.approx(ls["achieved"] / ls["peak"], abs=2e-4) and 0 < ls["frac"] < vi["frac"] < 1
    # HBM: 2 x FETCH + WRITE, KB -> bytes, per pass of the PROFILED shape; fractions of the 8 TB/s peak over ms_per_step
    traffic = (2 * 400000 + 300000) * 1024.0
    assert roof["hbm"]["traffic_bytes_per_pass"] == pytest.approx(traffic) and roof["hbm"]["traffic_frac"] == pytest.approx(traffic / 0.8e-3 / 8e12, abs=1e-4)
    assert roof["hbm"]["traffic_over_algorithmic"] == pytest.approx(traffic / 8.0e9, abs=1e-3)
    # the algorithmic bytes against HBM (> 1: cache-resident tree) and against the guide's L2 rate (< 1)
    assert roof["hbm"]["algorithmic_rate_over_peak"] == pytest.approx(8.0e9 / 0.8e-3 / 8e12, abs=1e-3) and roof["hbm"]["algorithmic_rate_over_peak"] > 1
    assert roof["hbm"]["algorithmic_rate_over_l2_peak"] == pytest.approx(8.0e9 / 0.8e-3 / 34.5e12, abs=1e-3) and roof["hbm"]["algorithmic_rate_over_l2_peak"] < 1
    assert roof["l2"]["hit_rate"] == pytest.approx(0.7) and roof["l1_accesses"]["achieved"] == pytest.approx(3e8 / 0.8e-3 / 1e9)
    assert roof["node_visits"]["achieved"] == pytest.approx(0.79e8 / 0.8e-3 / 1e9, abs=0.01)  # record visits, not box tests (two per visit)
    assert roof["node_visits"]["box_tests_per_s"] == pytest.approx(1.6e8 / 0.8e-3 / 1e9, abs=0.01)
    assert roof["node_visits"]["frac"] == pytest.approx(0.79e8 / 0.8e-3 / 1e9 / BL.STEP_PEAK_GVISITS, abs=0.001)
    # the dominant kernel's wave states, and one row per kernel family sorted by summed time
    assert roof["k_trace_wave_states"]["executing"] == pytest.approx(0.4) and roof["k_trace_wave_states"]["s_waitcnt"] == pytest.approx(0.45)
    assert roof["k_trace_wave_states"]["issue_wait"] == pytest.approx(0.15)
    ks = roof["kernels"]
    assert [k["kernel"] for k in ks] == ["k_trace", "k_shade", "k_gen"]
    assert ks[0]["ms_summed_per_pass"] == pytest.approx(3.0) and ks[0]["launches_per_pass"] == pytest.approx(2 / passes, abs=1e-3)
    assert ks[0]["share_of_summed_kernel_time"] == pytest.approx(3.0 / 3.6, abs=1e-3) and ks[1]["lane_util"] == pytest.approx(0.9)
    assert ks[1]["hbm_fetch_bytes_per_pass"] == pytest.approx(2 * 300000 * 1024.0) and ks[1]["hbm_write_bytes_per_pass"] == pytest.approx(250000 * 1024.0)
    assert ks[0]["valu_instr_per_pass"] == pytest.approx(5e8)
    # live HIP-event figures of the dominant kernel
    assert roof["kernel_avg_ms"] == pytest.approx(2.0) and roof["kernel_concurrency"] == pytest.approx(4.0)
    # what binds: the largest like-by-like fraction, named; and the contract's per-launch fraction under both observed schedules
    # (4e9 bytes per launch: 2.0 ms live -> 0.25; the trace child's launches average 1.5 ms x 60 -> 4e9 / 90 ms / 8e12)
    b = roof["binding"]
    assert set(b["candidates"]) == {"lane_slots", "l1_accesses", "node_visits", "l2_rate"}
    assert b["frac"] == max(v for v in b["candidates"].values() if v is not None) and b["candidates"][b["resource"]] == b["frac"]
    assert b["candidates"]["node_visits"] == roof["node_visits"]["frac"] and b["candidates"]["l2_rate"] == roof["hbm"]["algorithmic_rate_over_l2_peak"]
    pr = roof["per_launch_frac_range"]
    assert pr["timed_passes_hip_events"] == pytest.approx(0.25) and pr["rocprof_trace_child"] == pytest.approx(4.0e9 / (1.5e-3 * passes) / 8e12, abs=1e-4)
    assert pr["min"] == min(pr["timed_passes_hip_events"], pr["rocprof_trace_child"]) and pr["max"] == max(pr["timed_passes_hip_events"], pr["rocprof_trace_child"])
    # a different number of profiled passes changes every per-pass figure: the denominator is the children's own pass count
    half = BL.assemble_roofline(ms, passes // 2, prof, trace, executed={"nodes": 1.6e8, "steps": 0.79e8, "algorithmic_bytes": 8.0e9},
                                reference={"nodes": 2.8e8, "algorithmic_bytes": 15.0e9}, kernel_events=(400.0, 200, 0.1), passes_timed=100)
    assert half["valu_issue"]["instr_per_pass"] == pytest.approx(2 * 5.6e8) and half["hbm"]["traffic_bytes_per_pass"] == pytest.approx(2 * traffic)
    assert half["traffic"] == pytest.approx(roof["traffic"])  # per LAUNCH of the dominant kernel: its own launch count divides it, not the pass count


def test_every_ratio_divides_like_by_like(tmp_path):
    """Rounds 2-4 printed node_visits.frac = BOX TESTS per second / a peak in NODE VISITS per second (two box tests per visit): 0.87-0.90 that
    were 0.44-0.45. Every block with achieved / peak / frac now names the quantity both are counted in (BL.RATIO_QUANTITIES), and each numerator
    is pinned here to the input it must come from — by feeding inputs in which the candidates differ by factors no rounding explains."""
    passes = 10
    _write_counters(str(tmp_path / "valu"), [(KT, "SQ_INSTS_VALU", 3e8 * passes), (KT, "SQ_ACTIVE_INST_VALU", 12e8 * passes), (KT, "SQ_THREAD_CYCLES_VALU", 12e8 * passes * 32)])
    _write_counters(str(tmp_path / "tcp"), [(KT, "TCP_TOTAL_CACHE_ACCESSES_sum", 2e8 * passes), (KT, "TCP_TOTAL_ACCESSES_sum", 7e8 * passes)])
    _write_counters(str(tmp_path / "fetch"), [(KT, "FETCH_SIZE", 1000 * passes)])
    _write_trace(str(tmp_path / "trace"), [(KT, 0, 1_000_000)] * 5)
    prof = {t: BL.read_counters(str(tmp_path / t)) for t in ("valu", "tcp", "fetch")}
    ex = {"nodes": 7.0e8, "steps": 1.0e8, "algorithmic_bytes": 3.0e9}     # 7 box tests per visit: nothing real, so that a mix-up cannot hide
    roof = BL.assemble_roofline(1.0, passes, prof, BL.read_kernel_trace(str(tmp_path / "trace")), executed=ex, reference={"nodes": 9e8, "algorithmic_bytes": 5e9},
                                kernel_events=(30.0, 20, 0.01), passes_timed=10)
    blocks = {"": roof, "lane_slots": roof["lane_slots"], "valu_issue": roof["valu_issue"], "l1_accesses": roof["l1_accesses"], "node_visits": roof["node_visits"]}
    assert set(blocks) == set(BL.RATIO_QUANTITIES)
    for name, b in blocks.items():
        assert b["quantity"] == BL.RATIO_QUANTITIES[name], name
        assert b["frac"] == pytest.approx(b["achieved"] / b["peak"], rel=2e-3), name
    # no other block of the object carries a frac without being listed
    extra = [k for k, v in roof.items() if isinstance(v, dict) and "frac" in v and "peak" in v and k not in blocks]
    assert extra == ["hbm"] or extra == [], extra       # (hbm: rates over named peaks, each key says which: *_over_peak, *_over_l2_peak, traffic_frac)
    s = 1.0e-3
    assert roof["node_visits"]["achieved"] == pytest.approx(ex["steps"] / s / 1e9, rel=1e-3)            # visits, NOT box tests
    assert roof["node_visits"]["achieved"] != pytest.approx(ex["nodes"] / s / 1e9, rel=0.5)
    assert "visits" in roof["node_visits"]["quantity"] and "visits" in roof["node_visits"]["unit"]
    assert roof["node_visits"]["box_tests_per_s"] == pytest.approx(ex["nodes"] / s / 1e9, rel=1e-3)      # kept beside it, under its own name
    assert roof["valu_issue"]["achieved"] == pytest.approx(3e8 / s / 1e9, rel=1e-3)                      # instructions, not active cycles (4x) nor thread cycles
    assert roof["lane_slots"]["achieved"] == pytest.approx(3e8 * 64 * 0.5 / s / 1e12, rel=1e-3)          # instructions x 64 x lane utilisation (32 of 64)
    assert roof["l1_accesses"]["achieved"] == pytest.approx(2e8 / s / 1e9, rel=1e-3)                     # cache accesses after coalescing, not requests (3.5x)
    # top level: bytes of ONE launch / duration of ONE launch — 20 launches in 10 passes, 30 ms summed
    assert roof["achieved"] == pytest.approx((3.0e9 / 2) / 1.5e-3 / 1e9, rel=1e-3)
    assert roof["achieved"] != pytest.approx(3.0e9 / s / 1e9, rel=0.3)                                   # not bytes per PASS over wall time per pass
    assert roof["traffic"] == pytest.approx(2 * 1000 * passes * 1024.0 / 5)                               # FETCH doubled, KB, / the kernel's 5 profiled launches


def test_roofline_block_without_counters_keeps_the_contract_keys():
    roof = BL.assemble_roofline(1.0, 0, None, None, executed={"nodes": 1e8, "steps": 0.5e8, "algorithmic_bytes": 1e9}, reference={"nodes": 2e8, "algorithmic_bytes": 2e9},
                                kernel_events=(0.0, 0, 0.0))
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof
    assert roof["bound"] == "hbm" and roof["frac"] is None and roof["lane_slots"]["frac"] is None and roof["traffic"] is None and roof["kernels"] is None and roof["hbm"]["algorithmic_rate_over_peak"] == pytest.approx(0.125)


def test_bench_py_uses_the_shared_definitions():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "BL.assemble_roofline(" in src and "BL.VALUE_DEFINITION" in src and "BL.METRIC_VERSION" in src
    assert "--render-only" in src and "PROFILE_REPEATS" in src
    assert BL.METRIC_VERSION == 4


def _lines():
    out = []
    for rnd, names in (("r03", ["bench.json", "bench_driver_config.json", "bench_dragon871k.json", "bench_cfg2.json", "bench_4k.json"]),
                       ("r04", sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r04", "bench*.json"))
                              if "steps1" not in p)),  # (bench_steps1.json: --no-profile --no-cpu-baseline, one pass alone: no roofline counters)
                       ("r05", sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r05", "bench*.json")) if "steps1" not in p)),
                       ("r06", sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r06", "bench*.json")) if "steps1" not in p))):
        out += [(rnd, n) for n in names]
    return out


@pytest.mark.parametrize("rnd,name", _lines())
def test_committed_bench_lines_keep_the_contract(rnd, name):
    path = os.path.join(ROOT, "profiles", rnd, name)
    d = json.loads(open(path).read().strip().split("\n")[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and "workload" in d["config"] and "model" not in d["config"]
    # the headline is the work the timed mode executed; the reference-defined rate stands beside it and is larger
    assert d["value"] == d["mrays_executed_per_s"] and d["mrays_reference_defined_per_s"] > d["value"]
    assert abs(d["value"] - d["rays_executed_per_step"] / d["ms_per_step"] / 1e3) < 0.02 * d["value"]
    lo, hi = d["ms_per_step_spread"]
    assert d["repeats"] >= 5 and lo <= d["ms_per_step"] <= hi
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert r["traffic"] > 0 and 0 < r["l2"]["hit_rate"] < 1 and 0 < r["hbm"]["traffic_frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and "sample" in c
    if rnd >= "r05":  # the top level follows the contract's formula; the lane-slot share is a block beside it; the timed frames are checked in-run
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and d["frames_verified"] is True
        pl = r["per_launch"]
        assert abs(r["achieved"] - pl["algorithmic_bytes"] / (pl["avg_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
        ls = r["lane_slots"]
        assert abs(ls["frac"] - r["valu_issue"]["frac"] * r["valu_issue"]["lane_util"]) < 2e-3
        for name in BL.RATIO_QUANTITIES:
            assert (r if name == "" else r[name])["quantity"] == BL.RATIO_QUANTITIES[name]
    if rnd >= "r06":  # what binds, named; the contract's fraction under both observed schedules
        b = r["binding"]
        assert b["resource"] in b["candidates"] and b["frac"] == max(v for v in b["candidates"].values() if v is not None) and 0 < b["frac"] < 1
        pr = r["per_launch_frac_range"]
        assert pr["min"] <= r["frac"] <= pr["max"] and pr["rocprof_trace_child"] is not None
    if rnd >= "r04":  # round 4's additions
        assert d["metric_version"] == (3 if rnd == "r04" else 4) and "value_definition" in d
        if rnd == "r04":
            assert r["bound"] == "valu_lanes" and abs(r["frac"] - r["valu_issue"]["frac"] * r["valu_issue"]["lane_util"]) < 2e-3
        ws = r["k_trace_wave_states"]
        assert ws and 0.9 < ws["executing"] + ws["s_waitcnt"] + ws["issue_wait"] < 1.1
        fams = [k["kernel"] for k in r["kernels"]]
        assert "k_trace" in fams and "k_shade" in fams and all("valu_instr_per_pass" in k for k in r["kernels"] if k["kernel"] in ("k_trace", "k_shade"))
        assert r["hbm"]["algorithmic_rate_over_l2_peak"] < 1
