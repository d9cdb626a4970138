"""The part of bench.py that turns measurements into the `roofline` block of its JSON line — pure functions of numbers, no GPU, no
subprocess — so that tests/test_bench_line.py can drive it on stubbed counters: a definition that drifts (a numerator and a
denominator describing different schedules, a fraction of the wrong peak) fails a test instead of reaching the driver.

The top level follows the bench contract to the letter: bound "hbm", achieved = algorithmic bytes of one launch of the
dominant kernel / its average launch duration, peak 8 TB/s, traffic = PMC HBM bytes per launch. What really binds the path (DESIGN.md
section 4) — branchy scalar fp32 per ray on a cache-resident tree — is told by the blocks beside it: the share of the chip's fp32 LANE
slots that do work (`lane_slots`: VALU issue-slot share x the fraction of lanes active in an issued instruction; the top-level figure of
rounds 3-4), the issue-slot share itself, the wave-state split of the dominant kernel (executing / s_waitcnt / waiting to issue),
vector-L1 accesses, node visits against the micro-benchmarked step, HBM traffic against the algorithmic bytes, and per-kernel rows."""
import csv
import glob
import os
import re

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0           # aggregate L2 bandwidth, same guide ("L2 (per XCD)": 4 MiB per XCD, ~34.5 TB/s)
VALU_PEAK_GINSTR = 1228.8       # 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (same guide, "Wave scheduling")
VALU_MEASURED_GINSTR = 1058.0   # the highest issue rate tools/ubench reaches on the box: independent 4-byte v_add_f32, 128 between two branches, 8 waves per
                                # SIMD = 2.32 cycles (6 waves: 2.44; 8-byte v_fma_f32: 2.54-2.72; profiles/r02/ubench.txt)
VALU_SAME_MIX_GINSTR = 846.0    # the six-face box test (aabb_entry) on registers (87 VALU + 17 SALU per test: selects, dependent chains) at k_trace's 6 waves per SIMD;
                                # the quick answers that replaced it for all but a few boxes in 100 000 (box_quick.h, ~50 VALU, few selects) issue no slower
L1_PEAK_GACC = 1010.0           # the highest vector-L1 (TCP) cache-access rate tools/ubench reaches on the box with the product's own node fetch
                                # (profiles/r02/l1_access_calibration.txt); one access per cycle and CU would be 614.4
STEP_PEAK_GVISITS = 244.7       # 64 lanes x 3.824e9 wave-steps/s: tools/ubench k_step<0, quick>, the product's traversal step (record fetch + two quick box answers) with
                                # every lane on its own L1-resident record at 6 waves per SIMD (profiles/r04/ubench.txt; the fetch alone: 4.03e9; with the six face
                                # tests of rounds 1-3: 3.55-3.62e9 = the 227 of earlier lines). Lanes that share a record are cheaper, so `frac` can pass 1.

VALUE_DEFINITION = ("rays EXECUTED by the timed fast mode (closest-hit + Sun-shadow BVH queries it really performs, device-counted in an untimed "
                    "mode-4 replay of the same passes) / wall time of the K timed passes")
METRIC_VERSION = 4  # 1: reference-defined rays (rounds 1-2); 2: executed rays (round 3); 3: executed rays, nearest-child-first walks (round 4:
                    # same ray count as 2, fewer node visits per ray); 4 (round 5): executed rays, every walk in the reference's order again (the
                    # nearest-first walk is opt-in: unprovable on the reference's phantom hits) — `ms_per_step` is the figure that compares across versions

# One small set per rocprofv3 --pmc pass (a set that asks for more than the hardware collects at once aborts the profiler); TA_* and
# TCP stall counters are left out: rocprofiler refuses them on gfx950 (profiles/r03/pmc_ta_tcp_sets_abort.txt).
PMC_SETS = (
    ("wave", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]),
    ("valu", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"]),
    ("tcp", ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TOTAL_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"]),
    ("tcc", ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"]),
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
)


def pick_available(counters, avail_text):
    """The counters of a set that `rocprofv3 --list-avail` offers on this box (whole-word match), in order."""
    return [c for c in counters if re.search(r"\b%s\b" % re.escape(c), avail_text)] if avail_text else list(counters)


def kernel_family(name):
    """`void (anonymous namespace)::k_trace<false, 6>(gd::Scene, ...)` -> `k_trace`."""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").strip()
    n = n.split("(")[0].split("<")[0]
    return n.split("::")[-1]


def read_counters(directory):
    """Sums of every counter over all dispatches of a rocprofv3 output directory: ({counter: total}, {family: {counter: total}})."""
    tot, fam = {}, {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            v = float(row["Counter_Value"])
            c = row["Counter_Name"]
            tot[c] = tot.get(c, 0.0) + v
            f = fam.setdefault(kernel_family(row["Kernel_Name"]), {})
            f[c] = f.get(c, 0.0) + v
    return tot, fam


def read_kernel_trace(directory):
    """{family: (launches, summed ms)} of a rocprofv3 --kernel-trace output directory."""
    out = {}
    for path in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            f = kernel_family(row["Kernel_Name"])
            n, ms = out.get(f, (0, 0.0))
            out[f] = (n + 1, ms + (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    return out


def _r(x, n=4):
    return None if x is None else round(x, n)


def wave_states(c):
    """Split of the summed wave time of a kernel (SQ_WAVE_CYCLES) into executing an instruction / sitting in s_waitcnt / ready but
    waiting for an issue slot — the reading of profiles/r03/thin_wave_modes.txt 0: ACTIVE_INST_ANY, WAIT_ANY, WAIT_INST_ANY."""
    w = c.get("SQ_WAVE_CYCLES")
    if not w:
        return None
    return {"executing": _r(c.get("SQ_ACTIVE_INST_ANY", 0.0) / w), "s_waitcnt": _r(c.get("SQ_WAIT_ANY", 0.0) / w),
            "issue_wait": _r(c.get("SQ_WAIT_INST_ANY", 0.0) / w), "wave_cycles_per_pass": None}


# Every block of the roofline object that carries achieved / peak / frac, with the quantity BOTH are counted in: a ratio of two
# different quantities is a bug (rounds 2-4 divided box tests by a peak in node visits, two box tests per visit — their 0.87-0.90
# were 0.44-0.45). tests/test_bench_line.py::test_every_ratio_divides_like_by_like holds every block to its entry here.
RATIO_QUANTITIES = {
    "": "algorithmic bytes of one launch of the dominant kernel per second of that launch",
    "lane_slots": "fp32 lane-operations per second",
    "valu_issue": "wave64 VALU instructions per second",
    "l1_accesses": "vector-L1 cache accesses per second",
    "node_visits": "interior-node visits (record fetches) per second",
}


def finish_roofline(roof, bytes_per_launch, trace, dominant):
    """What a reader should take away, added once every block is filled in:
      * `binding`: the resource the pass comes CLOSEST to saturating — the largest of the like-by-like fractions of lane slots, vector-L1
        accesses, node visits and the aggregate L2 rate — with that fraction: the contract's top-level `frac` prices a cache-resident tree
        against HBM, which is not what limits it;
      * `per_launch_frac_range`: the contract's per-launch fraction under the two schedules this invocation observed — the timed passes (HIP
        events, live) and the un-instrumented rocprofv3 --kernel-trace child: the same bytes per launch over two average launch durations.
        How far they lie apart is how much the figure depends on how many launches of other pass lanes overlap a launch, i.e. how little it
        says about the kernel itself."""
    cands = {"lane_slots": roof["lane_slots"].get("frac"), "l1_accesses": roof["l1_accesses"].get("frac"), "node_visits": roof["node_visits"].get("frac"),
             "l2_rate": roof["hbm"].get("algorithmic_rate_over_l2_peak")}
    have = {k: v for k, v in cands.items() if v is not None}
    if have:
        top = max(have, key=lambda k: have[k])
        roof["binding"] = {"resource": top, "frac": have[top], "candidates": cands,
                           "quantity": RATIO_QUANTITIES.get(top, "algorithmic bytes per second of wall time against the guide's aggregate L2 rate"),
                           "note": "the largest of the fractions that divide like by like (each block says how it is measured); none is near 1: the "
                                   "dominant kernel is bound by dependent latency (its waves' s_waitcnt share: %s_wave_states), not by a throughput" % dominant}
    fr = {"timed_passes_hip_events": roof.get("frac"), "rocprof_trace_child": None}
    if bytes_per_launch and trace and dominant in trace and trace[dominant][0]:
        avg_ms = trace[dominant][1] / trace[dominant][0]
        fr["rocprof_trace_child"] = _r(bytes_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
        fr["rocprof_trace_avg_ms"] = _r(avg_ms, 5)
    vals = [v for v in (fr["timed_passes_hip_events"], fr["rocprof_trace_child"]) if v is not None]
    roof["per_launch_frac_range"] = {"min": min(vals) if vals else None, "max": max(vals) if vals else None, **fr,
                                     "note": "the contract's frac (algorithmic bytes of one launch / average launch duration / 8 TB/s) under the two schedules "
                                             "observed; a launch's duration depends on how many launches of other pass lanes run beside it (kernel_concurrency)"}


def assemble_roofline(ms_per_step, passes_profiled, prof, trace, executed, reference, kernel_events, dominant="k_trace", passes_timed=None):
    """The `roofline` object of the bench line.
      ms_per_step      : wall time of one timed pass (median repetition), ms
      passes_profiled  : passes every profiled child rendered (the counters' denominator: the child renders the TIMED shape —
                         K passes, repeated — and nothing else)
      prof             : {set tag: (totals, per family)} from read_counters, or None
      trace            : {family: (launches, summed ms)} of an un-instrumented --kernel-trace child, or None
      executed / reference : dicts with nodes (box tests), algorithmic_bytes per pass (device counters of the fast mode / of the reference's
                         work); executed also steps (interior-node visits = record fetches)
      kernel_events    : (summed ms, launches, timed seconds) of the dominant kernel's launches measured live with HIP events
      passes_timed     : passes rendered while those events were collected (K x repetitions)
    Top level = the contract of the task statement: bound "hbm"; achieved = ALGORITHMIC bytes one launch of the dominant
    kernel processes / that kernel's average launch duration (HIP events on its own stream, live); peak = 8 TB/s; traffic = HBM bytes
    per launch from the PMC counters. Beside it the views that say what really binds a cache-resident tree (`lane_slots`, `valu_issue`,
    `l1_accesses`, `node_visits`, wave states, per-kernel rows).
    Every fraction is achieved / peak of the SAME quantity (RATIO_QUANTITIES) over the SAME passes; None where a counter is missing."""
    s = ms_per_step * 1e-3
    n = float(max(1, passes_profiled))
    ev_ms, ev_n, ev_span = kernel_events
    launches_per_pass = (ev_n / float(passes_timed)) if (passes_timed and ev_n) else None
    avg_ms = ev_ms / ev_n if ev_n else None
    bytes_per_launch = executed["algorithmic_bytes"] / launches_per_pass if launches_per_pass else None
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if (bytes_per_launch and avg_ms) else None
    roof = {"bound": "hbm", "achieved": _r(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": _r(achieved / HBM_PEAK_GBS) if achieved else None,
            "traffic": None, "quantity": RATIO_QUANTITIES[""],
            "definition": "SURVEY.md 8(d) / the bench contract: algorithmic bytes (48 B per box tested + 32/48/64/80 B per primitive tested + 32 B per pixel "
                          "and pass, as the fast mode EXECUTES them: device counters of an untimed mode-4 replay of the timed passes) of ONE launch of the "
                          "dominant kernel (bytes per pass / its launches per pass) / that kernel's average launch duration, measured live with HIP events on "
                          "the stream it is launched on; peak = 8 TB/s HBM3E; traffic = HBM bytes per launch from rocprofv3 PMC (2 x FETCH_SIZE + WRITE_SIZE of "
                          "that kernel, KB -> bytes, / its launches). Reading it: the tree (8.7 MB) is served by L1 / L2 / Infinity Cache, so algorithmic "
                          "bytes are not HBM bytes (`traffic` is ~7x smaller) and up to 8 launches overlap (`kernel_concurrency`): the wall-clock "
                          "algorithmic rate is hbm.algorithmic_rate_executed, above the HBM peak. What binds the kernel is in lane_slots / node_visits / "
                          "%s_wave_states" % dominant,
            "per_launch": {"algorithmic_bytes": bytes_per_launch, "avg_ms": _r(avg_ms, 5), "launches_per_pass": _r(launches_per_pass, 3), "traffic_bytes": None},
            "lane_slots": {"achieved": None, "peak": round(VALU_PEAK_GINSTR * 64 / 1e3, 2), "unit": "T fp32 lane-operations/s", "frac": None,
                           "quantity": RATIO_QUANTITIES["lane_slots"],
                           "definition": "useful share of the chip's fp32 lane slots (the top-level figure of rounds 3-4): wave64 VALU instructions of every "
                                         "kernel of a pass / ms_per_step x the lanes active in them (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), against 1024 "
                                         "SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction x 64 lanes = valu_issue.frac x valu_issue.lane_util. A share of "
                                         "lane SLOTS spent on instructions, not a figure of merit: it falls when the same boxes are tested with fewer instructions"},
            "kernel": dominant + " (closest-hit + Sun-shadow BVH-query launches of the wavefront pipeline)",
            "kernel_avg_ms": _r(ev_ms / max(1, ev_n), 5), "kernel_launches": ev_n,
            "kernel_concurrency": _r(ev_ms / (ev_span * 1e3), 3) if ev_span else None,
            "kernel_avg_ms_note": "HIP events around every launch of that kernel on the stream it is launched on, during the timed passes; launches of "
                                  "up to 8 pass lanes overlap (`kernel_concurrency` = summed kernel time / wall time), so per-launch figures are diagnostics",
            "valu_issue": {"peak": VALU_PEAK_GINSTR, "unit": "G wave64 VALU instructions/s", "achieved": None, "frac": None, "quantity": RATIO_QUANTITIES["valu_issue"],
                           "peak_measured": VALU_MEASURED_GINSTR, "peak_same_instruction_mix": VALU_SAME_MIX_GINSTR,
                           "definition": "SQ_INSTS_VALU of every kernel of a pass / ms_per_step: a share of ISSUE SLOTS (wasted instructions raise it)"},
            "l1_accesses": {"peak": L1_PEAK_GACC, "unit": "G vector-L1 (TCP) cache accesses/s", "achieved": None, "frac": None, "quantity": RATIO_QUANTITIES["l1_accesses"],
                            "definition": "TCP_TOTAL_CACHE_ACCESSES_sum of every kernel of a pass / ms_per_step against the highest rate tools/ubench "
                                          "reaches on the box with the product's node-fetch shape (one access per cycle and CU: 614.4)"},
            "node_visits": {"peak": STEP_PEAK_GVISITS, "unit": "G record visits/s (lane level: one lane fetching one node record and testing its two boxes)",
                            "quantity": RATIO_QUANTITIES["node_visits"],
                            "achieved": _r(executed["steps"] / s / 1e9, 2), "frac": _r(executed["steps"] / s / 1e9 / STEP_PEAK_GVISITS),
                            "box_tests_per_s": _r(executed["nodes"] / s / 1e9, 2),
                            "definition": "interior-node visits the fast mode executes (device counter box_steps: one record fetch + two box tests each) / "
                                          "ms_per_step, against 64 lanes x the rate of tools/ubench's traversal step with every lane on its own L1-resident "
                                          "record at k_trace's occupancy (own peak: a second opinion). CORRECTED at the end of round 4: the lines of rounds "
                                          "2-4 divided BOX TESTS (two per visit; still here as box_tests_per_s) by this peak in VISITS — their 0.87-0.90 "
                                          "were 0.44-0.45. What the visit rate lacks to the peak: lanes idle in a wave's step (lane_util), leaf steps, "
                                          "refills and waits (`*_wave_states`)"},
            "hbm": {"peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "algorithmic_bytes_per_pass_reference": reference["algorithmic_bytes"],
                    "algorithmic_bytes_per_pass_executed": executed["algorithmic_bytes"],
                    "algorithmic_rate_executed": _r(executed["algorithmic_bytes"] / s / 1e9, 1),
                    "algorithmic_rate_over_peak": _r(executed["algorithmic_bytes"] / s / 1e9 / HBM_PEAK_GBS, 3),
                    "algorithmic_rate_over_l2_peak": _r(executed["algorithmic_bytes"] / s / 1e9 / L2_PEAK_GBS, 3),
                    "note": "algorithmic bytes (SURVEY.md 8(d): 48 B per box tested + 32/48/64/80 B per primitive tested + 32 B per pixel and "
                            "pass) / ms_per_step. Above the HBM peak because the tree is served by L1 / L2 / Infinity Cache, not HBM: against the "
                            "guide's aggregate L2 rate (34.5 TB/s) the same bytes are `algorithmic_rate_over_l2_peak`; what really crosses to "
                            "memory is `traffic_*` (PMC)"},
            "kernels": None}
    if not prof:
        finish_roofline(roof, bytes_per_launch, trace, dominant)
        return roof
    valu_t, valu_f = prof.get("valu", ({}, {}))
    wave_t, wave_f = prof.get("wave", ({}, {}))
    tcp_t, tcp_f = prof.get("tcp", ({}, {}))
    tcc_t, _ = prof.get("tcc", ({}, {}))
    fetch_t, fetch_f = prof.get("fetch", ({}, {}))
    write_t, write_f = prof.get("write", ({}, {}))
    vi = roof["valu_issue"]
    if valu_t.get("SQ_INSTS_VALU"):
        instr = valu_t["SQ_INSTS_VALU"] / n
        vi["instr_per_pass"] = instr
        vi["achieved"] = _r(instr / s / 1e9, 2)
        vi["frac"] = _r(instr / s / 1e9 / VALU_PEAK_GINSTR)
        vi["frac_of_measured_peak"] = _r(instr / s / 1e9 / VALU_MEASURED_GINSTR)
        vi["frac_of_same_mix_peak"] = _r(instr / s / 1e9 / VALU_SAME_MIX_GINSTR)
        vi["salu_instr_per_pass"] = valu_t.get("SQ_INSTS_SALU", 0.0) / n
        vi["vmem_read_instr_per_pass"] = valu_t.get("SQ_INSTS_VMEM_RD", 0.0) / n
        if valu_t.get("SQ_ACTIVE_INST_VALU"):
            lu = valu_t["SQ_THREAD_CYCLES_VALU"] / (64.0 * valu_t["SQ_ACTIVE_INST_VALU"])
            vi["lane_util"] = _r(lu)
            roof["lane_slots"]["achieved"] = _r(instr * 64 * lu / s / 1e12, 3)
            roof["lane_slots"]["frac"] = _r(instr * lu / s / 1e9 / VALU_PEAK_GINSTR)
    if tcp_t.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        acc = tcp_t["TCP_TOTAL_CACHE_ACCESSES_sum"] / n
        l1 = roof["l1_accesses"]
        l1.update({"l1_cache_accesses_per_pass": acc, "l1_requests_before_coalescing_per_pass": tcp_t.get("TCP_TOTAL_ACCESSES_sum", 0.0) / n,
                   "l1_misses_to_l2_per_pass": tcp_t.get("TCP_TCC_READ_REQ_sum", 0.0) / n,
                   "achieved": _r(acc / s / 1e9, 2), "frac": _r(acc / s / 1e9 / L1_PEAK_GACC), "frac_of_one_access_per_clock": _r(acc / s / 1e9 / 614.4)})
    if tcc_t.get("TCC_REQ_sum"):
        hit, miss = tcc_t.get("TCC_HIT_sum", 0.0), tcc_t.get("TCC_MISS_sum", 0.0)
        roof["l2"] = {"requests_per_pass": tcc_t["TCC_REQ_sum"] / n, "hits_per_pass": hit / n, "misses_per_pass": miss / n,
                      "hit_rate": _r(hit / max(1.0, hit + miss)),
                      "note": "TCC_HIT / TCC_MISS / TCC_REQ summed over the L2 channels and all kernels of a pass; misses go on to the Infinity Cache and HBM"}
    if "FETCH_SIZE" in fetch_t or "WRITE_SIZE" in write_t:
        # FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B; calibrated for wide streams — 16-B gathers are uncalibrated: an upper
        # estimate), both counters in KB; Infinity-Cache hits are counted too (the guide's HBM / rocprofv3 section)
        fetch, write = fetch_t.get("FETCH_SIZE", 0.0) / n, write_t.get("WRITE_SIZE", 0.0) / n
        traffic = (2.0 * fetch + write) * 1024.0
        # the contract's `traffic`: HBM bytes of ONE launch of the dominant kernel (its own counters / its launches in the profiled children)
        dom_launches = trace[dominant][0] if (trace and dominant in trace) else 0
        if dom_launches:
            dom_traffic = (2.0 * fetch_f.get(dominant, {}).get("FETCH_SIZE", 0.0) + write_f.get(dominant, {}).get("WRITE_SIZE", 0.0)) * 1024.0 / dom_launches
            roof["traffic"] = dom_traffic
            roof["per_launch"]["traffic_bytes"] = dom_traffic
        roof["hbm"].update({"traffic_bytes_per_pass": traffic, "FETCH_SIZE_KB_per_pass": fetch, "WRITE_SIZE_KB_per_pass": write,
                            "traffic_rate": _r(traffic / s / 1e9, 1), "traffic_frac": _r(traffic / s / 1e9 / HBM_PEAK_GBS),
                            "traffic_over_algorithmic": _r(traffic / max(1.0, executed["algorithmic_bytes"]), 3)})
    # per kernel family: where the time, the instructions and the memory traffic of a pass go
    fams = sorted(set(valu_f) | set(wave_f) | set(fetch_f) | set(write_f) | set(trace or {}))
    rows = []
    total_ms = sum(ms for _, ms in (trace or {}).values()) or None
    for f in fams:
        v, w = valu_f.get(f, {}), wave_f.get(f, {})
        row = {"kernel": f}
        if trace and f in trace:
            row["launches_per_pass"] = _r(trace[f][0] / n, 3)
            row["ms_summed_per_pass"] = _r(trace[f][1] / n)
            row["share_of_summed_kernel_time"] = _r(trace[f][1] / total_ms)
        if v.get("SQ_INSTS_VALU") is not None:
            row["valu_instr_per_pass"] = v.get("SQ_INSTS_VALU", 0.0) / n
            row["salu_instr_per_pass"] = v.get("SQ_INSTS_SALU", 0.0) / n
            if v.get("SQ_ACTIVE_INST_VALU"):
                row["lane_util"] = _r(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]))
        if f in fetch_f or f in write_f:
            row["hbm_fetch_bytes_per_pass"] = 2.0 * fetch_f.get(f, {}).get("FETCH_SIZE", 0.0) / n * 1024.0
            row["hbm_write_bytes_per_pass"] = write_f.get(f, {}).get("WRITE_SIZE", 0.0) / n * 1024.0
        if f in tcp_f:
            row["l1_cache_accesses_per_pass"] = tcp_f[f].get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / n
        ws = wave_states(w)
        if ws:
            ws["wave_cycles_per_pass"] = w["SQ_WAVE_CYCLES"] / n
            row["wave_states"] = ws
        rows.append(row)
    rows.sort(key=lambda r: -(r.get("ms_summed_per_pass") or 0.0))
    roof["kernels"] = rows
    finish_roofline(roof, bytes_per_launch, trace, dominant)
    dom = next((r for r in rows if r["kernel"] == dominant), None)
    roof[dominant + "_wave_states"] = dom.get("wave_states") if dom else None
    roof["kernels_note"] = ("per kernel family and pass: ms_summed = kernel durations of an un-instrumented --kernel-trace child (launches overlap: the "
                            "sum exceeds ms_per_step), instruction counts / lane_util / wave states / memory bytes from the --pmc children (which "
                            "serialise the kernels: counts are schedule-independent, the wave-state split is that of a launch running alone)")
    return roof
