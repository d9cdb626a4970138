"""The run planner of libgpuart_hip.so (csrc/hip/run_planner.h) driven without a GPU through gpuart_hip_test_planner:
random sequences of resize / share / plan / mode / pass / flush must only ever produce pipeline runs that fit a lane's
path buffers. Round 2 had a memory-access fault on record (gpurun_out/k20_plans.txt: a 10-pass run at 1080p handed to lanes
that hold 8 passes); the planner invariants below are what rules that shape out."""
import ctypes as C

import numpy as np
import pytest

from gpuart_amd import binding as B

RESIZE, SHARE, PLAN, MODE, PASS, FLUSH, ALLOC_FAILS = range(7)
DEFAULTS = dict(batch_limit=64, lanes=8, batch_mpaths=16, min_run_kpaths=2048, small_kpaths=8500, lane_budget_mb=16384, plan_percent=75)


def plan(ops, max_runs=4096, **cfg):
    """-> list of (op index, n_slots, max_batch, passes, k_run, pending afterwards)."""
    c = dict(DEFAULTS); c.update(cfg)
    cfgv = (C.c_uint32 * 8)(c["batch_limit"], c["lanes"], c["batch_mpaths"], c["min_run_kpaths"], c["small_kpaths"], c["lane_budget_mb"], c["plan_percent"], 0)
    flat = np.ascontiguousarray(np.array(ops, np.uint32).reshape(-1, 3))
    out = np.zeros((max_runs, 6), np.uint32)
    L = B.hip_lib()
    L.gpuart_hip_test_planner.restype = C.c_int
    n = L.gpuart_hip_test_planner(cfgv, flat.ctypes.data_as(C.POINTER(C.c_uint32)), len(flat), out.ctypes.data_as(C.POINTER(C.c_uint32)), max_runs)
    assert n >= 0, "gpuart_hip_test_planner -> %d (%s)" % (n, L.gpuart_hip_last_error().decode())
    assert n <= max_runs
    return [tuple(int(v) for v in row) for row in out[:n]]


def check(ops, runs, cfg):
    """Every run fits its lane; passes are launched in order, each exactly once; nothing is pending after a flush-like op."""
    c = dict(DEFAULTS); c.update(cfg)
    budget = c["batch_mpaths"] << 20
    requested = launched = 0
    by_op = {}
    for r in runs:
        by_op.setdefault(r[0], []).append(r)
    mode = 0
    have_tile = False
    for k, (op, a, b) in enumerate(ops):
        if op == PASS and mode != 2:
            requested += a
        if op == MODE:
            mode = a
        for (_, n_slots, max_batch, count, k_run, pending) in by_op.get(k, []):
            assert 1 <= count <= max_batch, "op %d %s: a run of %d passes in lanes that hold %d" % (k, ops[k], count, max_batch)
            assert n_slots * count <= max(budget, n_slots), "op %d: %d paths in one run" % (k, n_slots * count)
            assert max_batch <= c["batch_limit"]
            launched += count
        if op in (RESIZE, SHARE, MODE, FLUSH):
            # whatever was pending has been launched
            assert launched == requested, "op %d %s: %d passes requested, %d launched" % (k, ops[k], requested, launched)
        if op in (RESIZE, SHARE):
            have_tile = True
    assert have_tile
    return requested, launched


def test_the_shape_that_faulted_in_round_2():
    """1080p, plan 20, ten passes + flush twice: no run may exceed the 8 passes a lane holds (16M paths / 2.07M slots)."""
    for mode in (0, 3, 5):
        ops = [(RESIZE, 1920, 1080), (MODE, mode, 0), (PLAN, 20, 0), (PASS, 10, 0), (FLUSH, 0, 0), (PASS, 10, 0), (FLUSH, 0, 0)]
        runs = plan(ops)
        check(ops, runs, {})
        assert all(r[2] == 8 for r in runs) and sum(r[3] for r in runs) == 20
        assert max(r[3] for r in runs) <= 8
    # the driver's bench command on a 1/8 share: 25 passes as few k_run launches, none longer than the lanes
    ops = [(RESIZE, 1920, 1080), (SHARE, 3, 8), (PLAN, 25, 0), (PASS, 25, 0), (FLUSH, 0, 0)]
    runs = plan(ops)
    check(ops, runs, {})
    assert sum(r[3] for r in runs) == 25


def test_planner_defaults_match_the_design_notes():
    """1080p -> 8 passes per lane, 4K -> 2 (16M paths); one pass observed alone and short sequences go through k_run."""
    r = plan([(RESIZE, 1920, 1080), (PLAN, 1, 0), (PASS, 1, 0)])
    assert r == [(2, 2073600, 8, 1, 1, 0)]
    r = plan([(RESIZE, 1920, 1080), (PLAN, 3, 0), (PASS, 3, 0)])
    assert [x[3:5] for x in r] == [(3, 1)]
    # round 6: four passes of a 1080p frame are still one k_run launch, five go through the pipeline (profiles/r06/k_run_vs_pipeline_small_k.txt)
    r = plan([(RESIZE, 1920, 1080), (PLAN, 4, 0), (PASS, 4, 0)], small_kpaths=0)     # 0 = the library's own default
    assert [x[3:5] for x in r] == [(4, 1)]
    r = plan([(RESIZE, 1920, 1080), (PLAN, 5, 0), (PASS, 5, 0), (FLUSH, 0, 0)], small_kpaths=0)
    assert all(x[4] == 0 for x in r) and sum(x[3] for x in r) == 5
    r = plan([(RESIZE, 1920, 1080), (PLAN, 64, 0), (PASS, 64, 0), (FLUSH, 0, 0)])
    assert all(x[4] == 0 for x in r) and sum(x[3] for x in r) == 64 and max(x[3] for x in r) <= 8
    r = plan([(RESIZE, 3840, 2160), (PASS, 5, 0), (FLUSH, 0, 0)])
    assert [x[2] for x in r] == [2] * 5 and [x[3] for x in r] == [1] * 5
    # a device that refuses the first allocations: fewer lanes, then shorter runs, same invariants
    r = plan([(ALLOC_FAILS, 4, 0), (RESIZE, 1920, 1080), (PLAN, 20, 0), (PASS, 20, 0), (FLUSH, 0, 0)])
    assert max(x[3] for x in r) <= r[0][2] < 8 and sum(x[3] for x in r) == 20


def test_a_planned_sequence_is_cut_into_as_many_equal_runs_as_there_are_lanes():
    """run_planner.h plan(), the rule of round 5 (plan_percent = 0, the library's default; lane budget 32 GB -> 8 lanes at 1080p):
    ceil(K / 8) passes per run; beyond 64 passes (8 lanes x 8 passes) in rounds of 8 runs; a share's small passes stay at >= 2M paths."""
    new = dict(plan_percent=0, lane_budget_mb=32768)
    def lengths(K, frame=(1920, 1080), share=None):
        ops = [(RESIZE,) + frame] + ([(SHARE, 0, share)] if share else []) + [(MODE, 3, 0), (PLAN, K, 0), (PASS, K, 0), (FLUSH, 0, 0)]
        runs = plan(ops, **new)
        check(ops, runs, new)
        return [r[3] for r in runs]
    assert lengths(20) == [3, 3, 3, 3, 3, 3, 2]
    assert lengths(24) == [3] * 8 and lengths(32) == [4] * 8 and lengths(64) == [8] * 8
    assert lengths(25) == [4] * 6 + [1] and lengths(9) == [2, 2, 2, 2, 1]
    assert lengths(72) == [5] * 14 + [2] and lengths(128) == [8] * 16    # two rounds of 8 runs, never 9 runs of 8
    assert lengths(20, share=4) == [4] * 5                                  # 0.52M paths per pass: not below 2M paths (4 passes) a run
    # only the plan's TAIL stays one run: what a read-back in the middle of the sequence flushes, and passes nobody planned after the
    # sequence is complete, are spread over the lanes (run_planner.h on_flush)
    seq = lambda *ops: [r[3] for r in plan([(RESIZE, 1920, 1080), (MODE, 3, 0)] + list(ops), **new)]
    assert seq((PLAN, 64, 0), (PASS, 13, 0), (FLUSH, 0, 0)) == [8, 1, 1, 1, 2]                       # 5 pending, mid-sequence: four runs
    assert seq((PLAN, 64, 0), (PASS, 61, 0), (FLUSH, 0, 0))[-5:] == [8, 1, 1, 1, 2]                    # (still mid-sequence)
    assert seq((PLAN, 13, 0), (PASS, 13, 0), (FLUSH, 0, 0)) == [2] * 6 + [1]                          # the tail: one run
    assert seq((PLAN, 69, 0), (PASS, 69, 0), (FLUSH, 0, 0))[-2:] == [5, 4]                            # the tail of 69 = 13 x 5 + 4: one run of 4
    assert seq((PLAN, 69, 0), (PASS, 69, 0), (FLUSH, 0, 0), (PASS, 4, 0), (FLUSH, 0, 0))[-4:] == [4, 1, 1, 2]  # 4 passes nobody planned
    assert seq((PLAN, 69, 0), (PASS, 69, 0), (FLUSH, 0, 0), (PASS, 69, 0), (FLUSH, 0, 0))[-2:] == [5, 4]     # the same sequence again
    # the rule of rounds 2-4 is still there for A/B (GPUART_HIP_PLAN_RUN_PERCENT=75)
    ops = [(RESIZE, 1920, 1080), (MODE, 3, 0), (PLAN, 24, 0), (PASS, 24, 0), (FLUSH, 0, 0)]
    assert [r[3] for r in plan(ops, plan_percent=75, lane_budget_mb=32768)] == [3] * 8
    old = [r[3] for r in plan([(RESIZE, 1920, 1080), (MODE, 3, 0), (PLAN, 64, 0), (PASS, 64, 0), (FLUSH, 0, 0)], plan_percent=75)]
    assert old[:10] == [6] * 10 and sum(old) == 64


@pytest.mark.parametrize("seed", range(6))
def test_random_sequences_never_overrun_a_lane(seed):
    rng = np.random.RandomState(1000 + seed)
    frames = [(64, 48), (256, 256), (1000, 7), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320), (123, 457), (8, 8), (4096, 4096)]
    for _ in range(60):
        cfg = {}
        if rng.rand() < 0.6:
            cfg = dict(batch_limit=int(rng.choice([1, 2, 5, 8, 64])), lanes=int(rng.choice([1, 2, 3, 8, 32])),
                       batch_mpaths=int(rng.choice([1, 4, 16, 64])), min_run_kpaths=int(rng.choice([64, 512, 2048, 8192])),
                       small_kpaths=int(rng.choice([0xffffffff, 64, 6400, 100000])), lane_budget_mb=int(rng.choice([64, 1024, 16384])),
                       plan_percent=int(rng.choice([0, 0, 10, 75, 300])))
        W, H = frames[rng.randint(len(frames))]
        ops = [(RESIZE, W, H)]
        for _ in range(rng.randint(5, 60)):
            k = rng.rand()
            if k < 0.45:
                ops.append((PASS, int(rng.choice([1, 1, 2, 3, 7, 10, 20, 64, 200])), 0))
            elif k < 0.6:
                ops.append((PLAN, int(rng.choice([0, 1, 2, 3, 8, 20, 25, 64, 1024])), 0))
            elif k < 0.72:
                ops.append((FLUSH, 0, 0))
            elif k < 0.82:
                ops.append((MODE, int(rng.randint(0, 6)), 0))
            elif k < 0.9:
                n = int(rng.choice([2, 3, 8]))
                if H >= 8 * n:
                    ops.append((SHARE, int(rng.randint(n)), n))
            elif k < 0.95:
                W, H = frames[rng.randint(len(frames))]
                ops.append((RESIZE, W, H))
            else:
                ops.append((ALLOC_FAILS, int(rng.randint(1, 9)), 0))
        ops.append((FLUSH, 0, 0))
        try:
            runs = plan(ops, max_runs=20000, **cfg)
        except AssertionError as e:
            # the only legitimate refusal: ALLOC_FAILS left nothing smaller to try
            assert "tile refused" in str(e), e
            continue
        req, got = check(ops, runs, cfg)
        assert req == got
