#!/usr/bin/env python3
"""Container-only soak: random cases (gpuart_amd.synth_scenes.random_case) rendered by the reference's GLSL on llvmpipe
and by the oracle, compared bit for bit; nothing is written.   python tests/golden/soak_oracle_vs_reference.py [first] [count]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as M  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle.glref import glref  # noqa: E402


WILD = "--wild" in sys.argv
if WILD:
    sys.argv.remove("--wild")
WILD2 = "--wild2" in sys.argv  # the second class of hostile numbers (gpuart_amd.synth_scenes.random_wild2_case)
if WILD2:
    sys.argv.remove("--wild2")
LATTICE = "--lattice" in sys.argv  # coplanar / coincident primitives on a coarse lattice: where the answer hinges on the reference's visiting order
if LATTICE:
    sys.argv.remove("--lattice")


GRAZING = "--grazing" in sys.argv  # frames full of rays that graze triangles: the phantom hits of shaders/triangle.glsl:50-76 (synth_scenes.random_grazing_case)
if GRAZING:
    sys.argv.remove("--grazing")


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    gl = glref.GLRef()
    progs, bad = {}, 0
    for seed in range(first, first + count):
        case = S.random_grazing_case(seed) if GRAZING else S.random_lattice_case(seed) if LATTICE else S.random_wild2_case(seed) if WILD2 else S.random_wild_case(seed) if WILD else S.random_case(seed)
        tree, _ = O.build_bvh(case["prims"])
        ms = case["max_segments"]
        if ms not in progs:
            progs[ms] = M.RefPrograms(gl, ms)
        r = M.RefRenderer(gl, progs[ms], case["W"], case["H"], case["cam"], tree)
        r.us, r.us_em, r.us_flags = case["user_sphere"], case["us_em"], case["us_flags"]
        r.sun_az, r.sun_alt, r.sun_on = case["sun_az"], case["sun_alt"], int(case["sun_on"])
        seeds = O.randseeds(case["passes"], seed=5489 + seed)
        ref_direct = r.direct()[..., :3].copy()
        r.reset()
        for k in range(case["passes"]):
            ref_acc = r.pt_pass(case["npaths"], seeds[k])
        W, H = case["W"], case["H"]
        sun = O.sun_direction(case["sun_az"], case["sun_alt"])
        P = O.make_params(sun, case["sun_alt"], case["sun_on"], case["user_sphere"], case["us_em"], case["us_flags"],
                          float(r.cam[12]), r.cam[0:3], ms, 0.01)
        d = O.render_direct(tree, r.cam, W, H, P)[0][..., :3]
        acc = np.zeros((H, W, 4), np.float32)
        for k in range(case["passes"]):
            O.pt_pass(tree, r.cam, W, H, P, seeds[k], case["npaths"], acc)
        for what, got, exp in (("direct", d, ref_direct), ("pt", acc[..., :3], ref_acc[..., :3])):
            same = (got.view(np.uint32) == exp.view(np.uint32)) | ((got == 0) & (exp == 0)) | (np.isnan(got) & np.isnan(exp))
            if not same.all():
                bad += 1
                print("seed %d %s: %d pixels differ (%d prims, flags %d, depth %d)" % (seed, what, int((~same.all(-1)).sum()),
                      len(case["prims"]), case["us_flags"], ms), flush=True)
        for t in (r.rstart, r.rdir, r.out, *r.acc):
            gl.L.glref_delete_tex(t)
    print("soak: %d cases, %d differing images" % (count, bad))


if __name__ == "__main__":
    main()
