#!/usr/bin/env python3
"""Looks for single RAYS on which the fast kernels' closest-hit walk (nearer child first + certificate, device_scene.h GD_NEAREST)
and the reference-order walk disagree, through the traversal test hook (gpuart_hip_test_traverse, any_hit 2 vs 0): camera rays,
then rounds of bounce rays off the hits (cosine-hemisphere directions from the device's own sampler, fresh seed per round) —
the population a path tracer produces. Disagreeing rays are printed and saved with the tree (npz) for a CPU post-mortem
(tests/order_debug.py). No oracle involved.   python3 tools/order_rays.py SCENE [rounds] [out.npz]"""
import os
os.environ.setdefault("GPUART_LIBDIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpuart_amd", "lib_test"))  # uses test hooks (include/gpuart_hip_test.h)
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

f32 = np.float32
scene = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
out = sys.argv[3] if len(sys.argv) > 3 else None
W, H = 1920, 1080
descs, camd = {"cfg3": (S.scene_d, S.BENCH_CAMERA), "cfg2": (S.scene_p, S.DEFAULT_CAMERA), "box": (S.box_scene, S.DEFAULT_CAMERA),
               "tree": (S.tree_scene, S.TREE_NEAR_CAMERA), "cluster": (S.cluster_scene, S.CLUSTER_NEAR_CAMERA),
               "lattice": (S.lattice_scene, S.DEFAULT_CAMERA)}[scene]
descs = descs()
cam = dict(camd); cam["dir"] = S.camera_dir(cam)
tree, depth = B.compile_bvh(descs)
be = B.Backend(0)
be.resize(W, H); be.upload_bvh(tree); be.set_camera(B.camera_basis(cam["pos"], cam["dir"], cam["up"], cam["fov_y"], cam["screen_dist"], W, H))
us = (0.0, 0.0, 0.0, 0.0)
rs0, rd0 = be.test_cam_rays()
rs0 = rs0.reshape(-1, 4).copy(); rd0 = rd0.reshape(-1, 4).copy()
rs, rd = rs0.copy(), rd0.copy()
rng = np.random.RandomState(12345)
found = []
total = 0
for k in range(rounds):
    a0, a1 = be.test_traverse(rs, rd, us)
    b0, b1 = be.test_traverse(rs, rd, us, nearest_first=True)
    total += len(rs)
    ga, gb = np.concatenate([a0, a1], 1), np.concatenate([b0, b1], 1)
    bad = ~((ga.view(np.uint32) == gb.view(np.uint32)) | (np.isnan(ga) & np.isnan(gb))).all(1)
    for i in np.nonzero(bad)[0]:
        print("round %d ray %d o %s d %s\n   reference order: t %.9g P %s N %s type %g\n   nearest first  : t %.9g P %s N %s type %g"
              % (k, i, rs[i, :3].tolist(), rd[i, :3].tolist(), a0[i, 0], a0[i, 1:4], a1[i, :3], a1[i, 3], b0[i, 0], b0[i, 1:4], b1[i, :3], b1[i, 3]), flush=True)
        found.append(np.concatenate([rs[i], rd[i], a0[i], a1[i], b0[i], b1[i]]))
    # next round: bounce off the hits (reference-order results), misses restart at the camera
    hit = a1[:, 3] >= 0
    seed = rng.uniform(0, 1, 3).astype(f32)
    v = np.zeros_like(rs); v[:, :3] = a1[:, :3]
    ri = np.zeros_like(rs); ri[:, :3] = a0[:, 1:4] + seed
    nd = be.test_hemisphere(v, ri)
    nrs, nrd = rs0.copy(), rd0.copy()
    nrd[:, :3] += (rng.uniform(-1, 1, 3) * 1e-4).astype(f32)  # another sub-pixel position
    nrs[hit, :3] = a0[hit, 1:4]; nrd[hit, :3] = nd[hit, :3]
    rs, rd = nrs, nrd
    if k % 10 == 9:
        print("round %d: %.3g rays so far, %d disagree" % (k, total, len(found)), flush=True)
print("%s: %.3g rays, %d on which the two walks disagree" % (scene, total, len(found)))
if out and found:
    np.savez_compressed(out, rays=np.array(found, f32), tree=tree, scene=scene)
be.close()
