#!/usr/bin/env python3
"""Randomised parity soak (GPU): random scenes of all four primitive types — including degenerate ones (zero radii,
zero-area and axis-aligned triangles, duplicates, cones with equal radii, coincident coplanar faces) — random cameras,
user-sphere modes, Sun on/off, path depths; direct lighting + a few path-tracing passes, compared bit for bit with the
oracle.   python3 tools/fuzz_parity.py [first_seed] [count]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

f32 = np.float32


def random_scene(rs):
    prims = []
    if rs.rand() < 0.8:
        prims.append((S.DISC, [0, 0, 0, 0, 0, 1, f32(rs.uniform(2, 6))]))
    n = int(rs.choice([0, 1, 2, 3, 8, 40, 200]))
    for _ in range(n):
        t = rs.randint(4)
        p = rs.uniform(-1.5, 1.5, 3); p[2] = abs(p[2])
        if t == S.SPHERE:
            r = rs.choice([0.0, rs.uniform(0.02, 0.4)], p=[0.05, 0.95])
            prims.append((S.SPHERE, [f32(p[0]), f32(p[1]), f32(p[2]), f32(r)]))
        elif t == S.DISC:
            nrm = rs.normal(size=3); nrm /= np.linalg.norm(nrm)
            if rs.rand() < 0.3:
                nrm = np.eye(3)[rs.randint(3)] * rs.choice([-1, 1])
            prims.append((S.DISC, [f32(p[0]), f32(p[1]), f32(p[2]), f32(nrm[0]), f32(nrm[1]), f32(nrm[2]), f32(rs.uniform(0, 0.5))]))
        elif t == S.TRIANGLE:
            a = p; b = p + rs.uniform(-0.5, 0.5, 3); c = p + rs.uniform(-0.5, 0.5, 3)
            mode = rs.rand()
            if mode < 0.15:   # axis-aligned flat triangle
                k = rs.randint(3); b[k] = a[k]; c[k] = a[k]
            elif mode < 0.2:  # zero area
                c = b.copy()
            tri = (S.TRIANGLE, [f32(x) for x in np.concatenate([a, b, c])])
            prims.append(tri)
            if rs.rand() < 0.1:
                prims.append(tri)  # exact duplicate: equal hit parameters, the first one must win
        else:
            q = p + rs.uniform(-0.4, 0.4, 3)
            r1, r2 = rs.uniform(0.0, 0.25, 2)
            if rs.rand() < 0.2:
                r2 = r1
            prims.append((S.CONE, [f32(p[0]), f32(p[1]), f32(p[2]), f32(q[0]), f32(q[1]), f32(q[2]), f32(r1), f32(r2)]))
    return prims


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    be = B.Backend(0)
    bad = 0
    for seed in range(first, first + count):
        rs = np.random.RandomState(seed)
        prims = random_scene(rs)
        W, H = int(rs.choice([17, 40, 64, 96])), int(rs.choice([9, 24, 48]))
        pos = rs.uniform(-2.5, 2.5, 3); pos[2] = abs(pos[2]) + 0.05
        target = rs.uniform(-0.5, 0.5, 3); target[2] = abs(target[2])
        d = target - pos; d /= np.linalg.norm(d)
        cam = O.camera(pos.astype(f32), d.astype(f32), (0.0, 0.0, 1.0), float(rs.uniform(30, 90)), 0.2, W, H)
        tree, _ = O.build_bvh(prims) if prims else O.build_bvh([])
        flags = int(rs.choice([0, 0, 1, 2, 6]))
        us = (float(rs.uniform(-1, 1)), float(rs.uniform(-1, 1)), float(rs.uniform(0.2, 1)), float(rs.choice([0.0, 0.25])))
        em = 3.0 if flags & 1 else 0.0
        sun = O.sun_direction(float(rs.uniform(0, 6.28)), float(rs.uniform(0.1, 1.5)))
        alt = float(rs.uniform(0.1, 1.5))
        P = O.make_params(sun, alt, bool(rs.rand() < 0.8), us, em, flags, float(cam[12]), cam[0:3],
                          int(rs.choice([1, 3, 5, 8])), 0.01)
        K, npaths = 3, int(rs.choice([1, 1, 2]))
        seeds = O.randseeds(K, seed=5489 + seed)
        exp_direct, _ = O.render_direct(tree, cam, W, H, P)
        acc = np.zeros((H, W, 4), f32)
        for k in range(K):
            O.pt_pass(tree, cam, W, H, P, seeds[k], npaths, acc)
        gp = B.Params(); C.memmove(C.byref(gp), C.byref(P), C.sizeof(gp))
        be.resize(W, H); be.upload_bvh(tree); be.set_camera(cam)
        res = {}
        for mode in (0, 2):
            be.set_mode(mode)
            be.render_direct(gp); res["direct", mode] = be.read(0)
            be.pt_reset(); be.pt_plan(K)
            for k in range(K):
                be.pt_pass(gp, seeds[k], npaths)
            res["pt", mode] = be.read(1)
        be.set_mode(0)
        ok = True
        for (what, mode), got in res.items():
            exp = exp_direct if what == "direct" else acc
            same = (got[..., :3].view(np.uint32) == exp[..., :3].view(np.uint32)) | ((got[..., :3] == 0) & (exp[..., :3] == 0)) \
                | (np.isnan(got[..., :3]) & np.isnan(exp[..., :3]))
            if not same.all():
                ok = False
                print("seed %d: %s mode %d: %d of %d pixels differ (%d prims, %dx%d, flags %d)"
                      % (seed, what, mode, int((~same.all(-1)).sum()), W * H, len(prims), W, H, flags), flush=True)
        bad += 0 if ok else 1
    be.close()
    print("fuzz: %d scenes, %d with differences" % (count, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
