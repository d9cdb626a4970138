// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product.
// C ABI over the CPU restatement, for ctypes (tests/, __graft_entry__.smoke(), bench.py's
// cpu_baseline leg). Arrays are (n,4) float32 unless stated otherwise.
#include <xmmintrin.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <algorithm>
#include <vector>

#include "device.h"
#include "host.h"

using namespace orc;

namespace {
inline V3 in3(const float *a, int i) { return V3{a[4 * i], a[4 * i + 1], a[4 * i + 2]}; }
inline V4 in4(const float *a, int i) { return V4{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]}; }
inline void out4(float *o, int i, float a, float b, float c, float d) {
    o[4 * i] = a; o[4 * i + 1] = b; o[4 * i + 2] = c; o[4 * i + 3] = d;
}

// llvmpipe runs shaders with denormals flushed (FTZ + DAZ): `1e-40f * 2` is 0 there [probed].
// Every entry point and worker thread of the restatement runs under the same MXCSR mode.
struct FtzDaz {
    unsigned saved;
    FtzDaz() : saved(_mm_getcsr()) { _mm_setcsr(saved | 0x8040u); }
    ~FtzDaz() { _mm_setcsr(saved); }
};

template <class F>
void parallel_rows(int rows, int nthreads, F f) {
    if (nthreads <= 1) { FtzDaz g; for (int r = 0; r < rows; r++) f(r, 0); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t] { FtzDaz g; for (int r; (r = next.fetch_add(1)) < rows;) f(r, t); });
    for (auto &t : th) t.join();
}
}  // namespace

extern "C" {

struct orc_stats {
    uint64_t rays, iterations, nodes, prim_tests[4], segments;
};

// ---- per-function batch evaluators ---------------------------------------------------------
void orc_random(const float *in, int n, float *out) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) {
        V4 v = in4(in, i);
        out4(out, i, random1(v.x), random2(v.x, v.y), random3(V3{v.x, v.y, v.z}), random4(v));
    }
}
void orc_sincos(const float *in, int n, float *out) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) { float s, c; sincos_lp(in[4 * i], &s, &c); out4(out, i, s, c, 0, 0); }
}
void orc_pow16(const float *in, int n, float *out) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) out4(out, i, pow_lp(in[4 * i], 16.0f), 0, 0, 0);
}
void orc_hemisphere(const float *v, const float *ri, int n, float *out) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) { V3 r = GetRandomHemisphereDirection(in3(v, i), in3(ri, i)); out4(out, i, r.x, r.y, r.z, 0); }
}
void orc_inside_cone(const float *v, const float *nrm, const float *ri, float halfAngle, int n, float *out) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) {
        V3 r = GetRandomDirectionInsideCone(in3(v, i), in3(nrm, i), halfAngle, in3(ri, i));
        out4(out, i, r.x, r.y, r.z, 0);
    }
}
void orc_sky(const float *dir, const float *sunDirAlt, int n, float *out) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) { V3 r = GetSkyColor(in3(dir, i), sunDirAlt); out4(out, i, r.x, r.y, r.z, 0); }
}
// out0 = (pos, P), out1 = (N, 0); P,N zeroed on a miss (undefined in the reference).
static void store_hit(float pos, V3 p, V3 nn, int i, float *o0, float *o1) {
    if (pos > 0) { out4(o0, i, pos, p.x, p.y, p.z); out4(o1, i, nn.x, nn.y, nn.z, 0); }
    else { out4(o0, i, pos, 0, 0, 0); out4(o1, i, 0, 0, 0, 0); }
}
void orc_sphere(const float *rs, const float *rd, const float *sph, int n, float *o0, float *o1) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) {
        float pos; V3 p{0, 0, 0}, nn{0, 0, 0}; V4 s = in4(sph, i);
        SphereIntersection(in3(rs, i), in3(rd, i), V3{s.x, s.y, s.z}, s.w, pos, p, nn);
        store_hit(pos, p, nn, i, o0, o1);
    }
}
void orc_disc(const float *rs, const float *rd, const float *cr, const float *dn, int n, float *o0, float *o1) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) {
        float pos; V3 p{0, 0, 0}, nn{0, 0, 0}; V4 c = in4(cr, i);
        DiscIntersection(in3(rs, i), in3(rd, i), V3{c.x, c.y, c.z}, c.w, in3(dn, i), pos, p, nn);
        store_hit(pos, p, nn, i, o0, o1);
    }
}
void orc_triangle(const float *rs, const float *rd, const float *v0, const float *v1, const float *v2, int n, float *o0,
                  float *o1) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) {
        float pos; V3 p{0, 0, 0}, nn{0, 0, 0};
        TriangleIntersection(in3(rs, i), in3(rd, i), in3(v0, i), in3(v1, i), in3(v2, i), pos, p, nn);
        store_hit(pos, p, nn, i, o0, o1);
    }
}
void orc_cone(const float *rs, const float *rd, const float *q0, const float *q1, const float *q2, const float *q3, int n,
              float *o0, float *o1) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) {
        float pos; V3 p{0, 0, 0}, nn{0, 0, 0}; V4 prm = in4(q3, i);
        ConeIntersection(in3(rs, i), in3(rd, i), in4(q0, i), in4(q1, i), in4(q2, i), prm.x, prm.y, prm.z, pos, p, nn);
        store_hit(pos, p, nn, i, o0, o1);
    }
}
/// AABB test against a 2-quad box per sample; out = (hit?1:0, pos)
void orc_aabb(const float *rs, const float *rd, const float *bmin, const float *bmax, int n, float *out) {
    FtzDaz ftz_guard;
    for (int i = 0; i < n; i++) {
        float box[8] = {bmin[4 * i], bmin[4 * i + 1], bmin[4 * i + 2], 0, bmax[4 * i], bmax[4 * i + 1], bmax[4 * i + 2], 0};
        V3 d = in3(rd, i); V3 rdiv{1 / d.x, 1 / d.y, 1 / d.z};
        float pos; bool h = IntersectsAABB(in3(rs, i), d, rdiv, box, 0, pos);
        out4(out, i, h ? 1.0f : 0.0f, h ? pos : 0.0f, 0, 0);
    }
}
/// Closest-hit query (optionally incl. user sphere); o0=(pos,P) o1=(N,type) ; type -1 on miss.
void orc_traverse(const float *tree, const float *rs, const float *rd, const float *userSphere, int n, float *o0,
                  float *o1, orc_stats *stats) {
    FtzDaz ftz_guard;
    TravStats st;
    for (int i = 0; i < n; i++) {
        Hit h; h.p = V3{0, 0, 0}; h.n = V3{0, 0, 0};
        bool ush = false;
        if (userSphere) CheckIntersectionInclUserSphere(in3(rs, i), in3(rd, i), tree, userSphere, h, ush, &st);
        else CheckBVHIntersection(in3(rs, i), in3(rd, i), tree, h, &st);
        if (h.ptype >= 0) { out4(o0, i, h.pos, h.p.x, h.p.y, h.p.z); out4(o1, i, h.n.x, h.n.y, h.n.z, (float)h.ptype + (ush ? 0.5f : 0)); }
        else { out4(o0, i, -1, 0, 0, 0); out4(o1, i, 0, 0, 0, -1); }
    }
    if (stats) {
        stats->rays = st.rays; stats->iterations = st.iterations; stats->nodes = st.nodes;
        for (int k = 0; k < 4; k++) stats->prim_tests[k] = st.prim_tests[k];
    }
}

// ---- host restatement ----------------------------------------------------------------------
/// prims: n records of {int type; float f[9]}. Returns malloc'd quad array (free with orc_free).
float *orc_build_bvh(const PrimDesc *prims, int n, unsigned maxLevels, unsigned minPrims, size_t *nquads, int *maxDepth) {
    std::vector<PrimRec> recs;
    recs.reserve(n);
    for (int i = 0; i < n; i++) recs.push_back(MakePrim(prims[i]));
    std::vector<float> t = BuildAndCompileBVH(std::move(recs), maxLevels, minPrims, maxDepth);
    float *o = (float *)malloc(t.size() * 4);
    memcpy(o, t.data(), t.size() * 4);
    *nquads = t.size() / 4;
    return o;
}
void orc_free(void *p) { free(p); }

/// std::sort as the reference calls it (src/bvh.cpp:96): elements compared by centre = 0.5 * (float sum) in double.
/// perm[i] = original position of the element left at i. (Checker for the product's parallel exact_sort.h.)
void orc_sort_permutation(const float *keys, size_t n, uint32_t *perm) {
    struct E { float k; uint32_t i; };
    std::vector<E> v(n);
    for (size_t i = 0; i < n; i++) v[i] = E{keys[i], (uint32_t)i};
    std::sort(v.begin(), v.end(), [](const E &a, const E &b) { return a.k * 0.5 < b.k * 0.5; });
    for (size_t i = 0; i < n; i++) perm[i] = v[i].i;
}

/// out[16] = pos(3) bl(3) dh(3) dv(3) pixelSize
void orc_camera(const float pos[3], const float dir[3], const float up[3], float fovY, float screenDist, unsigned W,
                unsigned H, float *out) {
    CameraBasis b = CameraScreenBasis(pos, dir, up, fovY, screenDist, W, H);
    memcpy(out, b.pos, 12); memcpy(out + 3, b.bottomLeft, 12); memcpy(out + 6, b.deltaHorz, 12); memcpy(out + 9, b.deltaVert, 12);
    out[12] = PixelSize(fovY, screenDist, H);
}
void orc_sun_direction(float az, float alt, float out[3]) { SunDirection(az, alt, out); }
/// First `npasses` RandSeed quadruples of a default-seeded (or `seed`) mt19937.
void orc_randseeds(uint32_t seed, int npasses, float *out) {
    RandSeedGen g;
    g.gen.seed(seed);
    for (int i = 0; i < npasses; i++) g.next(out + 4 * i);
}

// ---- frames ----------------------------------------------------------------------------------
/// cam = 12 floats from orc_camera. Image rows bottom-up (row 0 = bottom), RGBA32F, tile-local.
void orc_cam_rays(const float *cam, int W, int H, float *rstart, float *rdir) {
    FtzDaz ftz_guard;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            V3 s, d;
            CamInitPixel(x, y, W, H, cam, cam + 3, cam + 6, cam + 9, s, d);
            out4(rstart, y * W + x, s.x, s.y, s.z, 0);
            out4(rdir, y * W + x, d.x, d.y, d.z, 0);
        }
}

/// The ray of path j's first segment of every pixel (path_tracing.glsl:141-175: the camera ray, jittered): what the path tracer's first
/// closest-hit query of a pass is asked — for tests that aim such a ray at something (tests/golden/make_golden.py order_adversary_frames).
void orc_first_segment_rays(const float *cam, int W, int H, const Params *P, const float *randSeed, int j, float *rstart, float *rdir) {
    FtzDaz ftz_guard;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            V3 s0, d0, s, d;
            CamInitPixel(x, y, W, H, cam, cam + 3, cam + 6, cam + 9, s0, d0);
            FirstSegmentRay(s0, d0, *P, randSeed, j, s, d);
            out4(rstart, y * W + x, s.x, s.y, s.z, 0);
            out4(rdir, y * W + x, d.x, d.y, d.z, 0);
        }
}

/// UV of selected pixels: xy = n x (int x, int y); out = n x 2 floats.
void orc_pixel_uv(const int *xy, int n, int W, int H, float *out) {
    FtzDaz ftz_guard;
    float coef[12];
    QuadUVCoefs(W, H, coef);
    for (int i = 0; i < n; i++) PixelUV(xy[2 * i], xy[2 * i + 1], W, H, coef, out[2 * i], out[2 * i + 1]);
}

static void merge_stats(orc_stats *dst, const std::vector<TravStats> &ts, const std::vector<uint64_t> &segs) {
    if (!dst) return;
    memset(dst, 0, sizeof *dst);
    for (size_t i = 0; i < ts.size(); i++) {
        dst->rays += ts[i].rays; dst->iterations += ts[i].iterations; dst->nodes += ts[i].nodes;
        for (int k = 0; k < 4; k++) dst->prim_tests[k] += ts[i].prim_tests[k];
        dst->segments += segs[i];
    }
}

void orc_render_direct(const float *tree, const float *cam, int W, int H, int x0, int y0, int tw, int th,
                       const Params *P, float *out_rgba, int nthreads, orc_stats *stats) {
    if (nthreads < 1) nthreads = 1;
    std::vector<TravStats> ts(nthreads);
    std::vector<uint64_t> segs(nthreads, 0);
    parallel_rows(th, nthreads, [&](int r, int t) {
        for (int c = 0; c < tw; c++) {
            V3 s, d;
            CamInitPixel(x0 + c, y0 + r, W, H, cam, cam + 3, cam + 6, cam + 9, s, d);
            V3 col = DirectLightingPixel(s, d, tree, *P, &ts[t]);
            out4(out_rgba, r * tw + c, col.x, col.y, col.z, 1.0f);
        }
    });
    merge_stats(stats, ts, segs);
}

/// One progressive pass: accum[tile] += sum of npaths paths (path_tracing.glsl:255).
void orc_pt_pass(const float *tree, const float *cam, int W, int H, int x0, int y0, int tw, int th, const Params *P,
                 const float *randSeed, int npaths, float *accum_rgba, int nthreads, orc_stats *stats) {
    if (nthreads < 1) nthreads = 1;
    std::vector<TravStats> ts(nthreads);
    std::vector<uint64_t> segs(nthreads, 0);
    parallel_rows(th, nthreads, [&](int r, int t) {
        for (int c = 0; c < tw; c++) {
            V3 s, d;
            CamInitPixel(x0 + c, y0 + r, W, H, cam, cam + 3, cam + 6, cam + 9, s, d);
            V3 col = PathTracingPixel(s, d, tree, *P, randSeed, npaths, &ts[t], &segs[t]);
            float *a = accum_rgba + 4 * (r * tw + c);
            a[0] = a[0] + col.x; a[1] = a[1] + col.y; a[2] = a[2] + col.z;
        }
    });
    merge_stats(stats, ts, segs);
}

}  // extern "C"
