#!/usr/bin/env python3
"""Generates tests/golden/host_math.npz from the reference's OWN compiled Vec3 arithmetic (container only).

oracle/_ref/libmathref.so is oracle/mathref/mathref.cpp compiled against /root/reference/src/math_types.h where it lies
(`make -C oracle mathref`): the reference's Vec3<float>/Vec3<double> methods, the expression of Renderer::GetSunDirection
(src/renderer.h:175-179) and the screen-basis / PixelSize statements of Renderer::SetCamera (src/renderer.cpp:139-161,573-574)
evaluated on that class. Inputs are seeded; inputs and outputs are stored, so the tests need neither the reference nor this
script.     python3 tests/golden/make_host_golden.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libmathref.so"))
p = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(20261004)
PI_F = np.float32(3.1415926)

# ---- Sun direction: (azimuth, altitude) -> vec3 -------------------------------------------------------------------
sun_in = np.concatenate([
    np.array([[PI_F, PI_F / np.float32(4)], [0, 0], [0.3, 0.1], [5.0, 1.5], [PI_F * 2, PI_F / 2], [-1.0, -0.5]], np.float32),
    np.stack([rng.uniform(0, 2 * np.pi, 250), rng.uniform(0, np.pi / 2, 250)], 1).astype(np.float32)])
sun_out = np.zeros((len(sun_in), 3), np.float32)
for i, (az, alt) in enumerate(sun_in):
    L.mref_sun_direction(C.c_float(az), C.c_float(alt), p(sun_out[i]))

# ---- camera: pos, dir, up, fovY, screenDist, W, H -> bottomLeft, deltaHorz, deltaVert, pixelSize -----------------
cams = []
for pos, W, H in (((0.1, -3.05, 1.0), 1920, 1080), ((0.1, -3.05, 1.0), 256, 256), ((0.1, -1.6, 0.9), 1920, 1080), ((0.1, -1.6, 0.9), 3840, 2160),
                  ((0.1, -1.6, 0.9), 7680, 4320), ((0.1, -3.05, 1.0), 37, 23)):
    d = np.float32([0, 0, 0.95]) - np.float32(pos)  # src/main.cpp:609-613
    cams.append(list(pos) + d.tolist() + [0, 0, 1, 60.0, 0.2, W, H])
for _ in range(250):
    pos = rng.uniform(-5, 5, 3); d = rng.normal(size=3) * rng.uniform(0.1, 4); up = rng.normal(size=3)
    cams.append(pos.tolist() + d.tolist() + up.tolist() + [rng.uniform(10, 120), rng.uniform(0.05, 2.0), int(rng.integers(8, 8000)), int(rng.integers(8, 5000))])
cam_in = np.array(cams, np.float32)  # W, H are exactly representable
cam_out = np.zeros((len(cam_in), 10), np.float32)
for i, c in enumerate(cam_in):
    L.mref_camera_basis(p(c[0:3].copy()), p(c[3:6].copy()), p(c[6:9].copy()), C.c_float(c[9]), C.c_float(c[10]), C.c_uint(int(c[11])), C.c_uint(int(c[12])),
                        p(cam_out[i]))

# ---- every Vec3 operation, float and double, incl. zero vectors, huge / tiny values ------------------------------
def vec_cases(n, dt):
    a = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3, (n, 1))
    b = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3, (n, 1))
    s = rng.normal(size=n) * 10.0 ** rng.uniform(-2, 2, n)
    a[0] = 0; b[1] = 0; s[2] = 0; a[3] = [1, 0, 0]; b[3] = [1, 0, 0]; a[4] = [1e30, -1e30, 1e-30]; s[5] = 1e-30
    return a.astype(dt), b.astype(dt), s.astype(dt)

fa, fb, fs = vec_cases(512, np.float32)
fo = np.zeros((512, 36), np.float32)
for i in range(512):
    L.mref_vec3f_ops(p(fa[i].copy()), p(fb[i].copy()), C.c_float(fs[i]), p(fo[i]))
da, db, ds = vec_cases(256, np.float64)
do = np.zeros((256, 36), np.float64)
for i in range(256):
    L.mref_vec3d_ops(p(da[i].copy()), p(db[i].copy()), C.c_double(ds[i]), p(do[i]))

# ---- cones: Cone::Cone's constants (double arithmetic) + world box, incl. cylinders, pointed, zero-length, negative and hostile values ----
nc = 1024
cc1 = rng.uniform(-2, 2, (nc, 3)).astype(np.float32)
cc2 = (cc1 + rng.normal(size=(nc, 3)) * rng.uniform(0.01, 2, (nc, 1))).astype(np.float32)
cr1 = rng.uniform(0.0, 0.6, nc).astype(np.float32)
cr2 = rng.uniform(0.0, 0.6, nc).astype(np.float32)
cr2[:128] = cr1[:128]                       # cylinders: CosB = 0
cr2[128:192] = 0                            # pointed
cr2[192:224] = cr1[192:224] + np.float32(5e-8)  # |r1 - r2| just below the 1e-7 cut
cc2[224:240] = cc1[224:240]                 # zero length: AxisLen = 0, the divisions give inf / NaN
hostile = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, -0.0, -0.3, 1e19, 3.4028234e38, 1e-45], np.float32)
for i in range(240, 400):
    arr = [cc1, cc2][int(rng.integers(2))] if rng.uniform() < 0.6 else None
    v = hostile[int(rng.integers(len(hostile)))]
    if arr is not None: arr[i, int(rng.integers(3))] = v
    elif rng.uniform() < 0.5: cr1[i] = v
    else: cr2[i] = v
cone_out = np.zeros((nc, 22), np.float32)
for i in range(nc):
    L.mref_cone(p(cc1[i].copy()), p(cc2[i].copy()), C.c_float(cr1[i]), C.c_float(cr2[i]), p(cone_out[i]))

np.savez_compressed(os.path.join(HERE, "host_math.npz"), cone_c1=cc1, cone_c2=cc2, cone_r1=cr1, cone_r2=cr2, cone_out=cone_out, sun_in=sun_in, sun_out=sun_out, cam_in=cam_in, cam_out=cam_out,
                    vec3f_a=fa, vec3f_b=fb, vec3f_s=fs, vec3f_out=fo, vec3d_a=da, vec3d_b=db, vec3d_s=ds, vec3d_out=do)
print("host_math.npz: %d sun directions, %d cameras, %d + %d Vec3 cases, %d cones" % (len(sun_in), len(cam_in), 512, 256, nc))
