"""Shared helpers for the parity tests: golden loading, scene lookup, bit-exact comparison."""
import os

import numpy as np

from gpuart_amd import synth_scenes as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def pad4(a, w=0.0):
    a = np.asarray(a, np.float32)
    if a.shape[-1] == 4:
        return a
    return np.concatenate([a, np.full(a.shape[:-1] + (4 - a.shape[-1],), w, np.float32)], -1)


SCENES = {
    "box": S.box_scene,
    "scene_p": S.scene_p,
    "scene_d": S.scene_d,
    "cluster": S.cluster_scene,
    "tree": S.tree_scene,
    "dragon871k": lambda: S.scene_d(660, 660),
    "scene_pc": lambda: S.scene_p(seed=3, nspheres=96, ndiscs=24, ncones=64),
    "lattice": S.lattice_scene,                               # coplanar / coincident primitives: the visiting order decides (round 4)
    "lattice_big": lambda: S.lattice_scene(seed=5, n=3000),   # ... on a tree large enough for the nearest-first kernels by default
}


def scene(name):
    return SCENES[name]()


def bit_mismatch(got, ref):
    """Number of rows whose float32 bit patterns differ (+0/-0 and NaN/NaN count as equal)."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    eq = (got.view(np.uint32) == ref.view(np.uint32)) | ((got == 0) & (ref == 0)) | (np.isnan(got) & np.isnan(ref))
    return int((~eq.reshape(eq.shape[0], -1).all(1)).sum()) if eq.ndim > 1 else int((~eq).sum())


def assert_bits(got, ref, what=""):
    n = bit_mismatch(got, ref)
    assert n == 0, "%s: %d of %d rows differ bitwise" % (what, n, np.asarray(ref).shape[0])


def rmse_per_channel(got, ref):
    d = np.asarray(got, np.float64)[..., :3] - np.asarray(ref, np.float64)[..., :3]
    return np.sqrt(np.nanmean(d * d, axis=tuple(range(d.ndim - 1))))


def frame_golden_params(O, g):
    """orc Params for a frames_*.npz fixture (defaults = reference defaults)."""
    cam = g["cam"]
    us = g["user_sphere"] if "user_sphere" in g else np.array(S.USER_SPHERE, np.float32)
    em = float(g["user_sphere_em"]) if "user_sphere_em" in g else 0.0
    fl = int(g["user_sphere_flags"]) if "user_sphere_flags" in g else 0
    ms = int(g["max_segments"]) if "max_segments" in g else 5
    sun = O.sun_direction(S.SUN_AZIMUTH, S.SUN_ALTITUDE)
    return lambda sun_on=True: O.make_params(sun, S.SUN_ALTITUDE, sun_on, us, em, fl, float(cam[12]), cam[0:3], ms, 0.01)


def fuzz_case_setup(O, seed):
    """(case, tree, camera, params, seeds) of random case `seed` (gpuart_amd.synth_scenes.random_case), as
    tests/golden/make_golden.py `fuzz` and tests/fuzz_parity.py set it up."""
    case = S.random_case(seed)
    cd = case["cam"]
    cam = O.camera(cd["pos"], cd["dir"], cd["up"], cd["fov_y"], cd["screen_dist"], case["W"], case["H"])
    tree, _ = O.build_bvh(case["prims"])
    sun = O.sun_direction(case["sun_az"], case["sun_alt"])
    P = O.make_params(sun, case["sun_alt"], case["sun_on"], case["user_sphere"], case["us_em"], case["us_flags"], float(cam[12]),
                      cam[0:3], case["max_segments"], 0.01)
    return case, tree, cam, P, O.randseeds(case["passes"], seed=5489 + seed)


def row_checksums(img):
    """Per row and channel: sum of the float32 bit patterns (uint64), as tests/golden/make_golden.py `fullsize` stores them."""
    return np.ascontiguousarray(img[..., :3], np.float32).view(np.uint32).astype(np.uint64).sum(axis=1)
