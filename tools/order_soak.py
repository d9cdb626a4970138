#!/usr/bin/env python3
"""Does the order in which the fast kernels visit a node's children (nearer child first, device_scene.h GD_NEAREST) ever change a
pixel? Two renderers on one GPU render the same passes of the same scene — one in the fast mode under test (0, 3 or 5), one in
mode 1 ("reference work": k_run in the reference's lower-then-upper order, every query in full) — and the accumulators are
compared bit for bit after every chunk of K passes. No oracle involved: the reference-order kernels are pinned to the reference by
the parity tests; this tool only asks whether the two orders agree, at sizes the oracle could not reach (1e9-1e11 rays).

  python3 tools/order_soak.py WORKLOAD [--frame WxH] [--passes K] [--chunks N] [--mode M]
  WORKLOAD: cfg3 | cfg2 | box | cluster | tree | dragon871k | lattice (coplanar, overlapping axis-aligned triangles: the case
            where a box is not conservative for its primitive in fp32 and the reference's own winner hinges on its visiting order)"""
import argparse
import os
os.environ.setdefault("GPUART_LIBDIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpuart_amd", "lib_test"))  # uses test hooks (include/gpuart_hip_test.h)
import sys
import tempfile
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("GPUART_HIP_NEAREST_MIN_PRIMS", "0")  # the nearest-first kernels also on trees the library would walk in the reference's order (small ones)
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402


def make(workload, W, H, mode, tmpdir, camera=None):
    cams = {"cfg3": S.BENCH_CAMERA, "dragon871k": S.BENCH_CAMERA, "cfg2": S.DEFAULT_CAMERA, "box": S.DEFAULT_CAMERA, "lattice": S.DEFAULT_CAMERA,
            "cluster": S.CLUSTER_NEAR_CAMERA, "tree": S.TREE_NEAR_CAMERA}
    cam = dict({"default": S.DEFAULT_CAMERA, "bench": S.BENCH_CAMERA, "cluster": S.CLUSTER_NEAR_CAMERA, "tree": S.TREE_NEAR_CAMERA}[camera] if camera else cams[workload])
    cam["dir"] = S.camera_dir(cam)
    r = B.Renderer(W, H, cam)
    r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
    segs = 5
    if workload in ("cfg3", "dragon871k"):
        r.set_primitives(B.make_prims(S.scene_d() if workload == "cfg3" else S.scene_d(660, 660))); segs = 8
    elif workload == "cfg2":
        r.set_primitives(B.make_prims(S.scene_p())); segs = 4
    elif workload == "box":
        r.init_box()
    elif workload == "lattice":
        r.set_primitives(B.make_prims(S.lattice_scene()))
    else:
        lines = S.cluster_dat_lines() if workload == "cluster" else S.tree_dat_lines()
        path = os.path.join(tmpdir, workload + ".dat")
        S.write_lines(path, lines)
        assert (r.init_cluster(path) if workload == "cluster" else r.init_tree(path))
    r.set_max_path_segments(segs)
    assert r.is_ok()
    r.backend.set_mode(mode)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--frame", default="1920x1080")
    ap.add_argument("--passes", type=int, default=64)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--camera", choices=["default", "bench", "cluster", "tree"], default=None, help="another camera than the workload's own")
    ap.add_argument("--ref-face-tests", action="store_true", help="the reference-order renderer runs every box through its six face tests "
                    "(GPUART_HIP_QUICK_BOXES=0 for its context): the fast mode's quick box answers (csrc/hip/box_quick.h) against IntersectsAABB "
                    "itself, over whole frames")
    a = ap.parse_args()
    W, H = (int(x) for x in a.frame.split("x"))
    with tempfile.TemporaryDirectory() as tmp:
        fast = make(a.workload, W, H, a.mode, tmp, a.camera)
        if a.ref_face_tests:
            os.environ["GPUART_HIP_QUICK_BOXES"] = "0"  # read when the context is created
        ref = make(a.workload, W, H, 1, tmp, a.camera)
        os.environ.pop("GPUART_HIP_QUICK_BOXES", None)
        fast.render_direct(); ref.render_direct()
        d = int((fast.read_direct()[..., :3].view(np.uint32) != ref.read_direct()[..., :3].view(np.uint32)).any(-1).sum())
        print("%s %dx%d: direct lighting: %d differing pixels" % (a.workload, W, H, d), flush=True)
        total_bad, rays, t_fast, t_ref = 0, 0, 0.0, 0.0
        for c in range(a.chunks):
            out = []
            for r in (fast, ref):
                r.backend.counters(reset=True)
                r.restart_path_tracing(1, a.passes)
                t0 = time.perf_counter()
                for _ in range(a.passes):
                    r.path_tracing_pass()
                r.finish()
                dt = time.perf_counter() - t0
                if r is fast: t_fast += dt
                else: t_ref += dt
                out.append(r.read_radiance(False))
            rays += ref.backend.counters().rays
            bad = (out[0][..., :3].view(np.uint32) != out[1][..., :3].view(np.uint32)).any(-1)
            total_bad += int(bad.sum())
            msg = ""
            if bad.any():
                ys, xs = np.nonzero(bad)
                msg = "  first: pixel (%d, %d) fast %s reference-order %s" % (xs[0], ys[0], out[0][ys[0], xs[0], :3], out[1][ys[0], xs[0], :3])
            print("chunk %d: %d passes, %d differing pixels%s" % (c, a.passes, int(bad.sum()), msg), flush=True)
        print("%s%s %dx%d mode %d vs mode 1: %d passes, %.3g reference-defined rays, %d differing pixel-chunks; %.3f / %.3f ms per pass (fast / reference order)"
              % (a.workload, " (%s camera)" % a.camera if a.camera else "", W, H, a.mode, a.passes * a.chunks, rays, total_bad, t_fast / (a.passes * a.chunks) * 1e3, t_ref / (a.passes * a.chunks) * 1e3), flush=True)
        fast.close(); ref.close()
    return 1 if total_bad or d else 0


if __name__ == "__main__":
    sys.exit(main())
