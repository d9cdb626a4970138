"""The quick box answer of the BVH queries (gpuart_amd/csrc/hip/box_quick.h) on the CPU: tools/quick_box_check.cpp compiles the very
source the kernels use, runs it under FTZ / DAZ against the reference's IntersectsAABB in its comparison form
(reference shaders/bvh_intersection.glsl:229-354) on adversarial rays, and fails on the first standing answer that differs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "quick_box_check.cpp")
FLAGS = ["-O2", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-pthread"]


def build(tmp_path, name, extra=()):
    exe = str(tmp_path / name)
    subprocess.run(["g++", *FLAGS, *extra, "-o", exe, SRC], check=True)
    return exe


def test_standing_quick_answers_are_the_reference_answers(tmp_path):
    r = subprocess.run([build(tmp_path, "qbc"), "8", "4", "3"], capture_output=True, text=True)
    assert r.returncode == 0 and "32000000 boxes, 0 mismatches" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    # the soak means something: the answer stands for a fair share of every class
    for line in r.stdout.splitlines()[:8]:
        assert float(line.split("quick answer stands")[1].split("%")[0]) > 10.0, line  # (the hostile and zero-component classes withdraw most: ~15 %)


@pytest.mark.parametrize("cut", ["-DQBC_CS_SCALE=0.01f", "-DQBC_CS_SCALE=0.0f", "-DGQ_RHO=0.0f", "-DGQ_NO_TINY_ORIGIN_GUARD"])
def test_the_soak_has_teeth(tmp_path, cut):
    """With the slack constant cut to a hundredth (or to nothing), without the relative margin, or without the guard that keeps rays with a
    tiny non-zero origin component away from the sign-based inside answer, the same soak must find answers that stand and are wrong."""
    r = subprocess.run([build(tmp_path, "qbc_cut", [cut]), "4", "4", "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "MISMATCH" in r.stderr, r.stdout[-800:]
