#!/usr/bin/env python3
"""Tuning sweep of the persistent BVH-query kernel's scheduling knobs on the bench workload (cfg3).
Runs on the GPU box; every configuration must reproduce the first one's accumulator bit for bit.
    python tools/sweep.py "WAVES_PER_CU=8,16,20" "LEAF_LANES=12,32" ...   (cartesian product)
"""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

W, H, K, REPS = 1920, 1080, int(os.environ.get("SWEEP_K", "24")), 3


def run(cfg, prims, cam, mode=0, reps=1):
    """One renderer with the knobs of `cfg`: best ms/pass of `reps` sequences of K passes, and the accumulator."""
    for k, v in cfg.items():
        os.environ["GPUART_HIP_" + k] = str(v)
    r = B.Renderer(W, H, cam)
    for k in cfg:
        del os.environ["GPUART_HIP_" + k]
    r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
    r.set_primitives(prims)
    r.set_max_path_segments(8)
    r.backend.set_mode(mode)
    r.backend.set_timing(0)
    r.restart_path_tracing(1, 2)
    r.path_tracing_pass(); r.path_tracing_pass()
    r.finish()
    dt = 1e9
    for _ in range(reps):
        r.set_seed(5489)
        r.restart_path_tracing(1, K)
        t0 = time.perf_counter()
        for _ in range(K):
            r.path_tracing_pass()
        r.finish()
        dt = min(dt, (time.perf_counter() - t0) / K * 1e3)
    acc = r.read_radiance(False)
    r.close()
    return dt, acc


def main():
    axes = []
    for a in sys.argv[1:]:
        k, vs = a.split("=")
        axes.append([(k, int(v)) for v in vs.split(",")])
    cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
    prims = B.make_prims(S.scene_d())
    combos = [dict(c) for c in itertools.product(*axes)] if axes else [{}]
    best = [1e9] * len(combos)
    same = [True] * len(combos)
    ref = None
    # the configurations take turns, REPS rounds: clock ramps and neighbours on the box hit all of them alike
    for _ in range(REPS):
        for i, cfg in enumerate(combos):
            dt, acc = run(cfg, prims, cam)
            if ref is None:
                ref = acc
            same[i] = same[i] and bool((acc.view(np.uint32) == ref.view(np.uint32)).all())
            best[i] = min(best[i], dt)
    for i, cfg in enumerate(combos):
        print("%-70s %8.3f ms/pass  identical=%s" % (cfg, best[i], same[i]), flush=True)
    dt, acc = run({}, prims, cam, mode=2)
    print("%-70s %8.3f ms/pass  identical=%s" % ("megakernel (mode 2)", dt, bool((acc.view(np.uint32) == ref.view(np.uint32)).all())))


if __name__ == "__main__":
    main()
