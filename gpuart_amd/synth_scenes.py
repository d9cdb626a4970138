"""Seeded synthetic scenes for tests and bench (the reference's data.zip is absent).

Scenes are lists of primitive descriptions `(type, [floats])`:
    sphere   (0): cx cy cz r
    disc     (1): cx cy cz  nx ny nz  r
    triangle (2): v0 v1 v2                (9 floats)
    cone     (3): c1 (3) c2 (3) r1 r2
fed unchanged to the product (`Renderer::SetPrimitives` through the C API) and to the oracle.
All coordinates are float32 values, so both sides see identical inputs.
"""
import numpy as np

SPHERE, DISC, TRIANGLE, CONE = 0, 1, 2, 3

# Reference default camera / lighting (src/main.cpp:609-613,622; src/renderer.cpp:202-209)
DEFAULT_CAMERA = dict(pos=(0.1, -3.05, 1.0), up=(0.0, 0.0, 1.0), dir=None, fov_y=60.0, screen_dist=0.2)
# Benchmark camera: closer, so the mesh fills a large part of the frame.
BENCH_CAMERA = dict(pos=(0.1, -1.6, 0.9), up=(0.0, 0.0, 1.0), dir=None, fov_y=60.0, screen_dist=0.2)
SUN_AZIMUTH = np.float32(3.1415926)
SUN_ALTITUDE = np.float32(3.1415926) / np.float32(4)
USER_SPHERE = (-0.4, 0.0, 0.2, 0.0)


def camera_dir(cam):
    """Dir = (0,0,0.95) - Pos, in float32 like Vec3f (src/main.cpp:611); cameras with a "target" look at it instead."""
    p = np.array(cam["pos"], np.float32)
    return tuple((np.array(cam.get("target", (0, 0, 0.95)), np.float32) - p).tolist())


def f32(x):
    return float(np.float32(x))


def box_scene():
    """The reference's InitBox (src/scenes.cpp:49-68): 9 hard-coded primitives."""
    T = lambda *v: (TRIANGLE, [f32(x) for x in v])
    return [
        (SPHERE, [0, 0, f32(0.3), f32(0.3)]),
        (DISC, [0, 0, 0, 0, 0, 1, 6]),
        T(1, -1, 0, 1, 1, 0, 1, 1, 1),
        T(1, -1, 0, 1, 1, 1, 1, -1, 1),
        (CONE, [f32(0.5), f32(-0.7), 0, f32(0.5), f32(-0.7), f32(0.35), f32(0.2), f32(0.2)]),
        T(1, 1, 0, 1, 1, 1, -1, 1, 1),
        T(-1, 1, 1, -1, 1, 0, 1, 1, 0),
        T(-1, -1, 0, -1, 1, 0, -1, 1, 1),
        T(-1, -1, 0, -1, 1, 1, -1, -1, 1),
    ]


def scene_p(seed=1, nspheres=256, ndiscs=16, ncones=0):
    """Scene P, "primitives-only" (cfg1/cfg2): floor disc + spheres resting on it + tilted discs."""
    rs = np.random.RandomState(seed)
    prims = [(DISC, [0, 0, 0, 0, 0, 1, 6])]
    for _ in range(nspheres):
        x, y = rs.uniform(-2, 2, 2)
        r = rs.uniform(0.03, 0.15)
        prims.append((SPHERE, [f32(x), f32(y), f32(r), f32(r)]))
    for _ in range(ndiscs):
        x, y = rs.uniform(-2, 2, 2)
        z = rs.uniform(0.2, 1.0)
        r = rs.uniform(0.1, 0.3)
        n = rs.uniform(-1, 1, 3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        prims.append((DISC, [f32(x), f32(y), f32(z), f32(n[0]), f32(n[1]), f32(n[2]), f32(r)]))
    for _ in range(ncones):
        x, y = rs.uniform(-2, 2, 2)
        h = rs.uniform(0.1, 0.5)
        r1, r2 = rs.uniform(0.02, 0.12, 2)
        dx, dy = rs.uniform(-0.1, 0.1, 2)
        prims.append((CONE, [f32(x), f32(y), 0.0, f32(x + dx), f32(y + dy), f32(h), f32(r1), f32(r2)]))
    return prims


def dragon_class_mesh(nu=224, nv=224):
    """Model-space vertices (float32, extent ~0.2) and faces of the dragon-class stand-in mesh:
    an upright displaced torus, nu*nv*2 triangles (224x224 -> 100 352), meant to be loaded
    like the reference's dragon: magnification 10, translation (0,0,-0.5) (src/scenes.cpp:35)."""
    u = (np.arange(nu) * (2 * np.pi / nu))[:, None]
    v = (np.arange(nv) * (2 * np.pi / nv))[None, :]
    R = 0.5
    r = 0.2 * (1.0 + 0.25 * np.sin(7 * u) * np.cos(5 * v) + 0.1 * np.sin(23 * v + 3 * u))
    wx = (R + r * np.cos(v)) * np.cos(u)
    wz = (R + r * np.cos(v)) * np.sin(u) + 0.85
    wy = r * np.sin(v) + 0 * u
    world = np.stack([wx, wy, wz], -1).reshape(-1, 3)
    model = ((world - np.array([0, 0, -0.5])) / 10.0).astype(np.float32)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j).ravel()
    b = (((i + 1) % nu) * nv + j).ravel()
    c = (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    d = (i * nv + (j + 1) % nv + 0 * i).ravel()
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int32)
    return model, faces


def load_transform(model, magnification=10.0, translation=(0.0, 0.0, -0.5)):
    """translation + magnification * v in float32, as LoadMeshFromPLY does (src/utils.cpp:110)."""
    t = np.array(translation, np.float32)
    return t[None, :] + np.float32(magnification) * model.astype(np.float32)


def scene_d(nu=224, nv=224):
    """Scene D, "dragon-class" (cfg3/4/5): the mesh loaded as InitDragon would + floor disc r=5."""
    model, faces = dragon_class_mesh(nu, nv)
    w = load_transform(model)
    tri = w[faces].reshape(-1, 9)
    prims = [(TRIANGLE, row.tolist()) for row in tri]
    prims.append((DISC, [0, 0, 0, 0, 0, 1, 5]))
    return prims


def write_ply(path, model, faces):
    """ASCII PLY in the dialect LoadMeshFromPLY accepts (src/utils.cpp:58-134)."""
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(model), len(faces)))
        for x, y, z in model:
            f.write("%.9g %.9g %.9g\n" % (x, y, z))
        for a, b, c in faces:
            f.write("3 %d %d %d\n" % (a, b, c))


def random_case(seed):
    """A random parity case (tests/fuzz_parity.py, tests/golden `fuzz`): primitives of all four types incl. degenerate
    ones (zero radii, zero-area and axis-aligned triangles, exact duplicates, cylinders), frame size, camera, user
    sphere, Sun, path depth, paths per pass. Everything is drawn from RandomState(seed) in a fixed order."""
    rs = np.random.RandomState(seed)
    prims = []
    if rs.rand() < 0.8:
        prims.append((DISC, [0, 0, 0, 0, 0, 1, f32(rs.uniform(2, 6))]))
    n = int(rs.choice([0, 1, 2, 3, 8, 40, 200]))
    for _ in range(n):
        t = rs.randint(4)
        p = rs.uniform(-1.5, 1.5, 3); p[2] = abs(p[2])
        if t == SPHERE:
            r = rs.choice([0.0, rs.uniform(0.02, 0.4)], p=[0.05, 0.95])
            prims.append((SPHERE, [f32(p[0]), f32(p[1]), f32(p[2]), f32(r)]))
        elif t == DISC:
            nrm = rs.normal(size=3); nrm /= np.linalg.norm(nrm)
            if rs.rand() < 0.3:
                nrm = np.eye(3)[rs.randint(3)] * rs.choice([-1, 1])
            prims.append((DISC, [f32(p[0]), f32(p[1]), f32(p[2]), f32(nrm[0]), f32(nrm[1]), f32(nrm[2]), f32(rs.uniform(0, 0.5))]))
        elif t == TRIANGLE:
            a = p; b = p + rs.uniform(-0.5, 0.5, 3); c = p + rs.uniform(-0.5, 0.5, 3)
            mode = rs.rand()
            if mode < 0.15:   # axis-aligned flat triangle
                k = rs.randint(3); b[k] = a[k]; c[k] = a[k]
            elif mode < 0.2:  # zero area
                c = b.copy()
            tri = (TRIANGLE, [f32(x) for x in np.concatenate([a, b, c])])
            prims.append(tri)
            if rs.rand() < 0.1:
                prims.append(tri)  # exact duplicate: equal hit parameters, the first one must win
        else:
            q = p + rs.uniform(-0.4, 0.4, 3)
            r1, r2 = rs.uniform(0.0, 0.25, 2)
            if rs.rand() < 0.2:
                r2 = r1
            prims.append((CONE, [f32(p[0]), f32(p[1]), f32(p[2]), f32(q[0]), f32(q[1]), f32(q[2]), f32(r1), f32(r2)]))
    W, H = int(rs.choice([17, 40, 64, 96])), int(rs.choice([9, 24, 48]))
    pos = rs.uniform(-2.5, 2.5, 3); pos[2] = abs(pos[2]) + 0.05
    target = rs.uniform(-0.5, 0.5, 3); target[2] = abs(target[2])
    d = target - pos; d /= np.linalg.norm(d)
    cam = dict(pos=tuple(float(f32(x)) for x in pos), dir=tuple(float(f32(x)) for x in d), up=(0.0, 0.0, 1.0),
               fov_y=float(f32(rs.uniform(30, 90))), screen_dist=0.2)
    flags = int(rs.choice([0, 0, 1, 2, 6]))
    us = (float(f32(rs.uniform(-1, 1))), float(f32(rs.uniform(-1, 1))), float(f32(rs.uniform(0.2, 1))), float(rs.choice([0.0, 0.25])))
    return dict(prims=prims, W=W, H=H, cam=cam, us_flags=flags, user_sphere=us, us_em=3.0 if flags & 1 else 0.0,
                sun_az=float(f32(rs.uniform(0, 6.28))), sun_alt=float(f32(rs.uniform(0.1, 1.5))), sun_on=bool(rs.rand() < 0.8),
                max_segments=int(rs.choice([1, 3, 5, 8])), npaths=int(rs.choice([1, 1, 2])), passes=3)


# second class of hostile numbers (--wild2): sign flips of ordinary values (negative radii), the reference's own magic
# numbers (closestPos / IntersectsAABB start at 1e19, VISIBILITY_OFFSET 1e-4, the 1e-8 / 1e-10 determinant cut-offs), the float
# range's ends, denormals (llvmpipe and the device flush them), and camera / user-sphere / Sun parameters out of the ordinary
WILD2 = [-0.3, -1.0, -0.05, 1e19, -1e19, 0.99e19, 1e-4, -1e-4, 1e-8, 1e-10, 3.4028234e38, -3.4028234e38, 1.17549435e-38,
         1e-39, 1e-45, -1e-45, 1e10, -1e10, 1e15, 0.5, 1.0]


def random_wild2_case(seed):
    case = random_case(seed)
    rs = np.random.RandomState(2000003 + seed)
    prims = []
    for t, vals in case["prims"]:
        vals = [f32(v) for v in vals]
        if rs.rand() < 0.3:
            for _ in range(int(rs.randint(1, 4))):
                k = int(rs.randint(len(vals)))
                vals[k] = f32(-vals[k]) if rs.rand() < 0.3 else f32(WILD2[int(rs.randint(len(WILD2)))])
        prims.append((t, vals))
    case["prims"] = prims
    if rs.rand() < 0.3:  # an odd camera: far away, on a primitive plane, looking straight along an axis, huge / tiny screen distance
        cam = dict(case["cam"])
        pick = int(rs.randint(5))
        if pick == 0: cam["pos"] = tuple(float(f32(x * 1e6)) for x in cam["pos"])
        elif pick == 1: cam["pos"] = (cam["pos"][0], cam["pos"][1], 0.0)
        elif pick == 2: cam["dir"] = (0.0, 0.0, -1.0); cam["up"] = (0.0, 1.0, 0.0)
        elif pick == 3: cam["screen_dist"] = float(f32(rs.choice([1e-6, 1e6, 1e19])))
        else: cam["fov_y"] = float(f32(rs.choice([0.001, 179.9, 1.0])))
        case["cam"] = cam
    if rs.rand() < 0.3:
        us = list(case["user_sphere"])
        us[int(rs.randint(4))] = float(f32(WILD2[int(rs.randint(len(WILD2)))]))
        case["user_sphere"] = tuple(us)
    if rs.rand() < 0.2:
        case["sun_alt"] = float(f32(rs.choice([0.0, -0.3, 1.5707964, 3.0, 1e-6])))
    return case


def random_wild_case(seed):
    """random_case(seed) with hostile numbers: some primitive coordinates replaced by NaN, +-inf, +-1e30, +-1e-30 or -0
    (tests/fuzz_parity.py --wild, tests/golden/soak_oracle_vs_reference.py --wild). The reference's comparisons decide what
    such primitives do; the product must decide the same."""
    case = random_case(seed)
    rs = np.random.RandomState(1000003 + seed)
    bad = [np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, -1e-30, -0.0, 0.0]
    prims = []
    for t, vals in case["prims"]:
        vals = [f32(v) for v in vals]
        if rs.rand() < 0.25:
            for _ in range(int(rs.randint(1, 3))):
                vals[int(rs.randint(len(vals)))] = f32(bad[int(rs.randint(len(bad)))])
        prims.append((t, vals))
    case["prims"] = prims
    return case


# third class (--lattice): geometry whose boxes are NOT conservative for their primitives in fp32 — coplanar, overlapping,
# axis-aligned triangles and discs with all coordinates on a coarse lattice (z-fighting surfaces), coincident spheres. A flat
# triangle's box has zero thickness, its entry parameter and the triangle's own hit parameter are the same number computed two
# ways, a few ulps apart in either direction: which of two coincident surfaces the reference reports then hinges on the
# order in which its walk meets them and on what it pruned in between. A walk in any other order has to notice (device_scene.h,
# GD_NEAREST) — this class is what checks that it does.
def lattice_prims(rs, n):
    L = lambda lo, hi: f32(rs.randint(int(lo * 4), int(hi * 4) + 1) / 4.0)
    prims = []
    for _ in range(n):
        kind = rs.rand()
        if kind < 0.75:  # axis-aligned triangle: all three vertices share coordinate k
            k = int(rs.randint(3))
            v = [[L(-1, 1), L(-1, 1), L(0, 1.25)] for _ in range(3)]
            c = L(0.25, 1.25) if k == 2 else L(-1, 1)
            for q in v:
                q[k] = c
            prims.append((TRIANGLE, [x for q in v for x in q]))
        elif kind < 0.9:  # axis-aligned disc on a lattice plane
            k = int(rs.randint(3))
            nrm = [0.0, 0.0, 0.0]; nrm[k] = float(rs.choice([-1, 1]))
            prims.append((DISC, [L(-1, 1), L(-1, 1), L(0.25, 1.25), f32(nrm[0]), f32(nrm[1]), f32(nrm[2]), f32(rs.choice([0.25, 0.5, 0.75]))]))
        else:  # spheres that coincide or touch
            prims.append((SPHERE, [L(-1, 1), L(-1, 1), L(0.25, 1.0), f32(rs.choice([0.25, 0.5]))]))
    return prims


def lattice_scene(seed=3, n=400):
    """A fixed scene of the lattice class for tools/order_soak.py (default camera)."""
    return [(DISC, [0, 0, 0, 0, 0, 1, 6])] + lattice_prims(np.random.RandomState(seed), n)


def random_lattice_case(seed):
    case = random_case(seed)
    rs = np.random.RandomState(3000017 + seed)
    case["prims"] = ([(DISC, [0, 0, 0, 0, 0, 1, 6])] if rs.rand() < 0.7 else []) + lattice_prims(rs, int(rs.choice([2, 3, 8, 40, 200])))
    return case


def random_grazing_case(seed):
    """Fifth fuzz class (round 5, `--grazing`): frames full of rays that GRAZE triangles — the regime in which the reference's
    Moeller-Trumbore (shaders/triangle.glsl:50-76) divides two cancelled sums and accepts phantom hits far from the triangle. A fan of
    triangulated sheets, each in a plane that (almost) contains the camera position — like the pages of a book seen from its spine —,
    spread over a narrow vertical field of view so that every pixel row runs along some sheet; any orientation (a third of the cases
    axis-aligned), a little off-plane noise per vertex; discs across the view and spheres stand between the sheets, so that there are
    surfaces between phantoms and their triangles' boxes. What the reference renders here is a property of its own visiting order
    (csrc/hip/device_scene.h): the class soaks the reference-order walks, and the oracle itself against the reference's GLSL
    (tests/golden/soak_oracle_vs_reference.py --grazing)."""
    case = random_case(seed)
    rs = np.random.RandomState(5000011 + seed)
    if rs.rand() < 0.33:
        k = rs.randint(3)
        n = np.eye(3)[k]; u = np.eye(3)[(k + 1) % 3]; v = np.eye(3)[(k + 2) % 3]
    else:
        n = rs.normal(size=3); n /= np.linalg.norm(n)
        u = np.cross(n, rs.normal(size=3)); u /= np.linalg.norm(u)
        v = np.cross(n, u)
    pos = rs.uniform(-0.5, 0.5, 3)
    fov = float(rs.choice([1.0, 3.0, 10.0, 40.0]))
    sheets = int(rs.choice([1, 4, 16, 40]))
    m = int(rs.choice([1, 2, 4]))
    size = float(rs.uniform(0.3, 1.5))
    noise = float(rs.choice([0.0, 1e-7, 1e-6, 1e-5]))
    prims = []
    for _ in range(sheets):
        th = np.radians(rs.uniform(-0.5, 0.5) * fov)
        uk = np.cos(th) * u + np.sin(th) * n            # the sheet's plane contains v and uk; its normal nk
        nk = np.cos(th) * n - np.sin(th) * u
        c0 = pos + uk * rs.uniform(0.3, 2.0) + nk * (float(rs.choice([0.0, 1e-6, 1e-5, 1e-4])) * float(rs.choice([-1, 1])))
        grid = np.zeros((m + 1, m + 1, 3))
        for i in range(m + 1):
            for j in range(m + 1):
                grid[i, j] = c0 + uk * (size * i / m) + v * (size * (2.0 * j / m - 1.0)) + nk * (noise * rs.uniform(-1, 1))
        for i in range(m):
            for j in range(m):
                a, b, cc, d = grid[i, j], grid[i + 1, j], grid[i + 1, j + 1], grid[i, j + 1]
                prims.append((TRIANGLE, [f32(x) for x in np.concatenate([a, b, cc])]))
                prims.append((TRIANGLE, [f32(x) for x in np.concatenate([a, cc, d])]))
    for _ in range(int(rs.choice([0, 2, 6, 16]))):   # discs across the view, spheres between the sheets
        th = np.radians(rs.uniform(-0.5, 0.5) * fov)
        q = pos + (np.cos(th) * u + np.sin(th) * n) * rs.uniform(0.3, 2.5) + v * rs.uniform(-size, size)
        if rs.rand() < 0.6:
            prims.append((DISC, [f32(q[0]), f32(q[1]), f32(q[2]), f32(u[0]), f32(u[1]), f32(u[2]), f32(rs.uniform(0.002, 0.1))]))
        else:
            prims.append((SPHERE, [f32(q[0]), f32(q[1]), f32(q[2]), f32(rs.uniform(0.002, 0.08))]))
    case["prims"] = prims
    case["cam"] = dict(pos=tuple(float(f32(x)) for x in pos), dir=tuple(float(f32(x)) for x in u), up=tuple(float(f32(x)) for x in n),
                       fov_y=float(f32(fov)), screen_dist=0.2)
    case["us_flags"], case["us_em"] = 0, 0.0
    case["user_sphere"] = (case["user_sphere"][0], case["user_sphere"][1], case["user_sphere"][2], 0.0)
    return case


# ---- stand-ins for the reference's two primitive-list scenes (data/cluster_100k.dat, data/tree1_21k.dat are absent) ----
# Written in the dialect Utils::LoadPrimitives reads (src/utils.cpp:136-203): `sphere x y z [r]` (r defaults to 4) and
# `cone x1 y1 z1 x2 y2 z2 r1 r2` lines, `#` comments; loaded as InitCluster / InitTree do (src/scenes.cpp:69-103).
CLUSTER_LOAD = dict(magnification=0.01, translation=(0.0, 0.0, 2.5))   # InitCluster, src/scenes.cpp:73
TREE_LOAD = dict(magnification=0.3, translation=(0.0, 0.0, 0.0))       # InitTree, src/scenes.cpp:91
FLOOR_DISC_CT = (DISC, [1, 0, 0, 0, 0, 1, 6])                          # both scenes add Disc((1,0,0),(0,0,1),6)


def _g9(x):
    return "%.9g" % float(np.float32(x))  # 9 significant digits identify a float32 uniquely


def cluster_dat_lines(n=100_000, seed=7):
    """A globular-cluster-like cloud of `n` spheres in model units (Plummer profile, core radius 40, cut at 200 so that
    after x0.01 and the lift by 2.5 everything floats above the floor); 5 % of the lines give no radius (-> 4)."""
    rs = np.random.RandomState(seed)
    u = rs.uniform(1e-4, 1.0, n)
    rad = np.minimum(40.0 / np.sqrt(u ** (-2.0 / 3.0) - 1.0 + 1e-12), 200.0)
    d = rs.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = d * rad[:, None]
    explicit = rs.uniform(size=n) < 0.95
    r = rs.uniform(0.4, 1.6, n)
    lines = ["# synthetic stand-in for cluster_100k.dat: %d spheres" % n]
    for i in range(n):
        s = "sphere %s %s %s" % (_g9(p[i, 0]), _g9(p[i, 1]), _g9(p[i, 2]))
        if explicit[i]:
            s += " " + _g9(r[i])
        lines.append(s)
    return lines


def tree_dat_lines(depth=8, seed=11):
    """A branching tree: every branch is a cone (tapering conical frustum) that spawns three thinner, shorter children;
    tips carry sphere "leaves". depth 8 -> 9 841 cones + ~11 000 spheres (the reference's tree1_21k: 21k cones + spheres)."""
    rs = np.random.RandomState(seed)
    lines = ["# synthetic stand-in for tree1_21k.dat"]

    def branch(base, direction, length, r0, level):
        tip = base + direction * length
        r1 = r0 * 0.7
        lines.append("cone %s %s" % (" ".join(_g9(v) for v in np.concatenate([base, tip])), _g9(r0) + " " + _g9(r1)))
        if level == depth:
            for _ in range(int(rs.randint(1, 3))):
                c = tip + rs.normal(size=3) * 0.05
                lines.append("sphere %s %s" % (" ".join(_g9(v) for v in c), _g9(rs.uniform(0.03, 0.09))))
            return
        for _ in range(3):
            nd = direction + rs.normal(size=3) * 0.55
            nd[2] = abs(nd[2]) * 0.6 + 0.15
            nd /= np.linalg.norm(nd)
            branch(tip, nd, length * 0.72, r1, level + 1)

    branch(np.zeros(3), np.array([0.0, 0.0, 1.0]), 2.2, 0.25, 0)
    return lines


def dat_descs(lines, magnification=1.0, translation=(0.0, 0.0, 0.0)):
    """What Utils::LoadPrimitives makes of the lines: translation + magnification * v and magnification * r in float32."""
    m = np.float32(magnification)
    t = np.array(translation, np.float32)
    out = []
    for ln in lines:
        tok = ln.split()
        if not tok or tok[0].startswith("#"):
            continue
        v = [np.float32(float(x)) for x in tok[1:]]
        if tok[0] == "sphere":
            c = t + m * np.array(v[:3], np.float32)
            r = m * (v[3] if len(v) > 3 else np.float32(4.0))
            out.append((SPHERE, [float(c[0]), float(c[1]), float(c[2]), float(r)]))
        elif tok[0] == "cone":
            c1 = t + m * np.array(v[0:3], np.float32)
            c2 = t + m * np.array(v[3:6], np.float32)
            out.append((CONE, [float(x) for x in c1] + [float(x) for x in c2] + [float(m * v[6]), float(m * v[7])]))
    return out


def cluster_scene(n=100_000, seed=7):
    """The primitive list InitCluster would hand to SetPrimitives for the synthetic cluster file."""
    return dat_descs(cluster_dat_lines(n, seed), **CLUSTER_LOAD) + [FLOOR_DISC_CT]


def tree_scene(depth=8, seed=11):
    """The primitive list InitTree would hand to SetPrimitives for the synthetic tree file."""
    return dat_descs(tree_dat_lines(depth, seed), **TREE_LOAD) + [FLOOR_DISC_CT]


def write_lines(path, lines):
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")

# A second camera for the cluster / tree scenes: above the floor, close to the crown / the cluster's core (the default
# camera sees both from 3 m away, mostly floor and sky).
CLUSTER_NEAR_CAMERA = dict(pos=(0.3, -2.6, 1.7), target=(0.0, 0.0, 2.3), up=(0.0, 0.0, 1.0), dir=None, fov_y=60.0, screen_dist=0.2)
TREE_NEAR_CAMERA = dict(pos=(0.6, -1.9, 1.9), target=(0.0, 0.0, 1.5), up=(0.0, 0.0, 1.0), dir=None, fov_y=60.0, screen_dist=0.2)
NEAR_CAMERAS = {"cluster": CLUSTER_NEAR_CAMERA, "tree": TREE_NEAR_CAMERA}
