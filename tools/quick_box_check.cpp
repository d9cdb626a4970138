// quick_box_check.cpp — soundness soak of gpuart_amd/csrc/hip/box_quick.h on the CPU: the very source the kernels compile, run
// under FTZ / DAZ (what the device and llvmpipe run) against the reference's IntersectsAABB in its comparison form
// (shaders/bvh_intersection.glsl:229-354, restated here statement by statement), on rays made to hurt: through corners and
// edges, origins on planes, flat / nested / tiny / huge boxes, dyadic coordinates (exact ties), zero, tiny, huge and
// non-finite components. Whenever the quick answer STANDS it must be the reference's.
//   g++ -O2 -mfma -ffp-contract=off -fno-fast-math -pthread -o /tmp/qbc tools/quick_box_check.cpp && /tmp/qbc [millions per thread] [threads] [seed]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <xmmintrin.h>
#include <pmmintrin.h>

static inline bool gq_nan(float a) { return a != a; }
static inline float gq_min(float a, float b) { return gq_nan(a) ? b : gq_nan(b) ? a : (a < b ? a : b); }  // v_min_f32: the operand that is not NaN
static inline float gq_max(float a, float b) { return gq_nan(a) ? b : gq_nan(b) ? a : (a > b ? a : b); }
static inline float gq_med3(float a, float b, float c) {  // v_med3_f32: min3 when an operand is NaN
    if (gq_nan(a) || gq_nan(b) || gq_nan(c)) return gq_min(gq_min(a, b), c);
    return gq_max(gq_min(a, b), gq_min(gq_max(a, b), c));
}
static inline float gq_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float gq_abs(float a) { return __builtin_fabsf(a); }
#define GQ_FN static inline
#include "../gpuart_amd/csrc/hip/box_quick.h"

struct V3 { float x, y, z; };

// ---- the reference, comparison form --------------------------------------------------------------------------------------
static bool ref_aabb(V3 o, V3 d, V3 rdiv, V3 lo, V3 hi, float &pos) {
    if (o.x >= lo.x && o.y >= lo.y && o.z >= lo.z && o.x <= hi.x && o.y <= hi.y && o.z <= hi.z) { pos = -1.0f; return true; }
    bool hit = false;
    float p = 1.0e+19f;
#define FACE(dc, plane, oc, rdivc, a0, a1, lo_a, hi_a, b0, b1, lo_b, hi_b)                  \
    if ((dc) != 0) {                                                                        \
        const float k = ((plane) - (oc)) * (rdivc);                                         \
        if (k >= 0) {                                                                       \
            const float a = (a0) + k * (a1), b = (b0) + k * (b1);                           \
            if (a >= (lo_a) && a <= (hi_a) && b >= (lo_b) && b <= (hi_b)) {                 \
                hit = true;                                                                 \
                if (k < p) p = k;                                                           \
            }                                                                               \
        }                                                                                   \
    }
    FACE(d.x, lo.x, o.x, rdiv.x, o.y, d.y, lo.y, hi.y, o.z, d.z, lo.z, hi.z)
    FACE(d.x, hi.x, o.x, rdiv.x, o.y, d.y, lo.y, hi.y, o.z, d.z, lo.z, hi.z)
    FACE(d.y, lo.y, o.y, rdiv.y, o.x, d.x, lo.x, hi.x, o.z, d.z, lo.z, hi.z)
    FACE(d.y, hi.y, o.y, rdiv.y, o.x, d.x, lo.x, hi.x, o.z, d.z, lo.z, hi.z)
    FACE(d.z, lo.z, o.z, rdiv.z, o.x, d.x, lo.x, hi.x, o.y, d.y, lo.y, hi.y)
    FACE(d.z, hi.z, o.z, rdiv.z, o.x, d.x, lo.x, hi.x, o.y, d.y, lo.y, hi.y)
#undef FACE
    pos = p;
    return hit;
}

// ---- the quick answer exactly as device_scene.h assembles it ---------------------------------------------------------------
static bool quick(V3 o, V3 d, V3 rdiv, V3 lo, V3 hi, float cs_tree, float &pos, bool &hit) {
    const float cs = gq_ray_slack(cs_tree, o.x, o.y, o.z, d.x, d.y, d.z, rdiv.x, rdiv.y, rdiv.z);
    const float k0 = (lo.x - o.x) * rdiv.x, k1 = (hi.x - o.x) * rdiv.x;
    const float k2 = (lo.y - o.y) * rdiv.y, k3 = (hi.y - o.y) * rdiv.y;
    const float k4 = (lo.z - o.z) * rdiv.z, k5 = (hi.z - o.z) * rdiv.z;
    return gq_box(k0, k1, k2, k3, k4, k5, gq_abs(rdiv.x), gq_abs(rdiv.y), gq_abs(rdiv.z), cs, pos, hit);
}

struct Rng {
    uint64_t s;
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    uint32_t u32() { return (uint32_t)(next() >> 24); }
    float uni() { return (float)(u32() & 0xffffff) * (1.0f / 16777216.0f); }          // [0,1)
    float sym() { return 2.0f * uni() - 1.0f; }
    int below(int n) { return (int)(u32() % (uint32_t)n); }
};

static float nudge(float v, int ulps) {  // moves v by `ulps` representable steps
    uint32_t b; memcpy(&b, &v, 4);
    int32_t i = (int32_t)b;
    i = i < 0 ? (int32_t)0x80000000 - i : i;
    i += ulps;
    i = i < 0 ? (int32_t)0x80000000 - i : i;
    b = (uint32_t)i; memcpy(&v, &b, 4);
    return v;
}
static float flushed(float v) { return std::fabs(v) < 1.17549435e-38f ? 0.0f * v : v; }

static const float SCALES[] = {1.0f, 1.0f, 5.0f, 5.0f, 0.01f, 1.0e-3f, 100.0f, 1.0e4f, 1048576.0f, 1.0e-20f};
static const float SPECIAL[] = {0.0f, -0.0f, 1.0e-39f, 1.0e30f, -1.0e30f, 3.0e38f, INFINITY, -INFINITY, NAN, 1.0e19f, 1.0e-30f, 1048576.0f, 2.0e6f,
                                -1.0e-37f, 2.0e-37f, -3.0e-35f, 1.0e-33f, -1.0e-20f};
static const int NSPECIAL = sizeof(SPECIAL) / sizeof(SPECIAL[0]);

struct Stats { uint64_t n = 0, sure = 0, sure_hit = 0, sure_in = 0, bad = 0, ref_hit = 0; };

static void worker(uint64_t seed, uint64_t iters, Stats *out) {
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
    Rng g{seed * 0x9E3779B97F4A7C15ull + 12345};
    Stats st[8];
    for (uint64_t it = 0; it < iters; it++) {
        const int cls = g.below(8);
        const float S = SCALES[g.below(10)];
        const bool dyadic = cls == 1 || (cls == 5);  // coordinates on a coarse binary grid: exact ties everywhere
        auto coord = [&](float s) { return dyadic ? s * (float)(g.below(33) - 16) * 0.125f : s * g.sym(); };
        V3 lo, hi;
        {
            float c[3], e[3];
            for (int k = 0; k < 3; k++) {
                c[k] = coord(S);
                const int m = g.below(10);
                e[k] = m == 0 ? 0.0f : m == 1 ? S * 1.0e-6f * g.uni() : m == 2 ? S * 1.0e-3f * g.uni() : dyadic ? S * (float)g.below(9) * 0.125f : S * g.uni() * (m < 6 ? 0.05f : 1.0f);
            }
            lo = V3{flushed(c[0] - e[0]), flushed(c[1] - e[1]), flushed(c[2] - e[2])};
            hi = V3{flushed(c[0] + e[0]), flushed(c[1] + e[1]), flushed(c[2] + e[2])};
            if (!(lo.x <= hi.x) || !(lo.y <= hi.y) || !(lo.z <= hi.z)) continue;
        }
        // a target point: corner, edge, face, interior or outside, possibly nudged by a few ulps
        float t[3];
        const int tk = g.below(6);
        for (int k = 0; k < 3; k++) {
            const float l = (&lo.x)[k], h = (&hi.x)[k];
            const int on = tk == 0 ? g.below(2) : tk == 1 ? (k == 2 ? 2 : g.below(2)) : tk == 2 ? (k == 0 ? g.below(2) : 2) : tk == 3 ? 2 : 3;
            t[k] = on == 0 ? l : on == 1 ? h : on == 2 ? l + (h - l) * g.uni() : l + (h - l) * (3.0f * g.uni() - 1.0f);
            if (tk == 5) t[k] += S * 0.5f * g.sym();
            if (g.below(3) == 0) t[k] = nudge(t[k], g.below(9) - 4);
        }
        V3 o, d;
        {
            float oo[3], dd[3];
            const int ok = g.below(8);
            const float far = ok < 2 ? 0.1f : ok < 4 ? 3.0f : ok < 6 ? 50.0f : 1.0e-3f;
            for (int k = 0; k < 3; k++) {
                oo[k] = dyadic ? coord(S) : t[k] + S * far * g.sym();
                if (ok == 7 && g.below(2)) oo[k] = g.below(2) ? (&lo.x)[k] : (&hi.x)[k];  // origin on a plane
                if (g.below(6) == 0) oo[k] = nudge(oo[k], g.below(9) - 4);
            }
            const float len = cls == 2 ? 1.0e-3f : cls == 3 ? 100.0f : 1.0f;
            float n2 = 0;
            for (int k = 0; k < 3; k++) { dd[k] = t[k] - oo[k]; n2 += dd[k] * dd[k]; }
            const float inv = (dyadic || !(n2 > 0)) ? 1.0f : len / std::sqrt(n2);
            for (int k = 0; k < 3; k++) {
                dd[k] *= inv;
                if (cls == 4 && g.below(3) == 0) dd[k] = (g.below(2) ? 1.0f : -1.0f) * (g.below(2) ? 6.2e-8f : 1.0e-12f) * (1.0f + g.uni());  // Sun-like
                if (cls == 6 && g.below(3) == 0) dd[k] = 0.0f * (g.below(2) ? 1.0f : -1.0f);
                if (g.below(8) == 0) dd[k] = nudge(dd[k], g.below(5) - 2);
            }
            if (cls == 5) {  // tiny origin components beside planes that are exactly 0, small direction components: the inside answer's flush corner
                for (int k = 0; k < 3; k++) {
                    if (g.below(3) == 0) oo[k] = (g.below(2) ? 1.0f : -1.0f) * (g.below(2) ? 1.0e-37f : 3.0e-33f) * (1.0f + g.uni());
                    if (g.below(3) == 0) dd[k] = (g.below(2) ? 1.0f : -1.0f) * 1.0e3f * (1.0f + g.uni());
                }
            }
            if (cls == 7) {  // hostile numbers
                for (int k = 0; k < 3; k++) {
                    if (g.below(4) == 0) oo[k] = SPECIAL[g.below(NSPECIAL)];
                    if (g.below(4) == 0) dd[k] = SPECIAL[g.below(NSPECIAL)];
                }
            }
            o = V3{oo[0], oo[1], oo[2]};
            d = V3{dd[0], dd[1], dd[2]};
        }
        const V3 rdiv{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
        float pmax = 0;
        for (int k = 0; k < 3; k++) pmax = std::fmax(pmax, std::fmax(std::fabs((&lo.x)[k]), std::fabs((&hi.x)[k])));
        if (g.below(4) == 0) pmax *= 1.0f + 10.0f * g.uni();  // (a tree's bound is looser than one box's)
        bool planes_ok = true;  // what the converter checks per tree (gq_plane_ok): no plane within 2^-60 of zero without being zero
        for (int k = 0; k < 3; k++) planes_ok = planes_ok && gq_plane_ok((&lo.x)[k]) && gq_plane_ok((&hi.x)[k]);
        if (!planes_ok) pmax = NAN;  // -> slack +inf
#ifdef QBC_CS_SCALE  // teeth test: a slack constant that is too small must produce mismatches
        const float cs = gq_slack_of_tree(pmax) * QBC_CS_SCALE;
#else
        const float cs = gq_slack_of_tree(pmax);
#endif
        float rp, qp;
        bool qh;
        const bool rh = ref_aabb(o, d, rdiv, lo, hi, rp);
        const bool sure = quick(o, d, rdiv, lo, hi, cs, qp, qh);
        Stats &s = st[cls];
        s.n++;
        s.ref_hit += rh;
        if (!sure) continue;
        s.sure++;
        s.sure_hit += qh;
        s.sure_in += qh && qp == -1.0f;
        const bool same = qh == rh && (!rh || qp == rp);  // (+0 == -0: entry parameters are only ever compared)
        if (!same) {
            if (s.bad++ < 3)
                fprintf(stderr, "MISMATCH class %d: o %.9g %.9g %.9g d %.9g %.9g %.9g lo %.9g %.9g %.9g hi %.9g %.9g %.9g cs %.9g  ref %d %.9g  quick %d %.9g\n", cls,
                        o.x, o.y, o.z, d.x, d.y, d.z, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, cs, rh, rp, qh, qp);
        }
    }
    for (int c = 0; c < 8; c++) { out[c].n += st[c].n; out[c].sure += st[c].sure; out[c].sure_hit += st[c].sure_hit; out[c].sure_in += st[c].sure_in; out[c].bad += st[c].bad; out[c].ref_hit += st[c].ref_hit; }
}

int main(int argc, char **argv) {
    const uint64_t millions = argc > 1 ? strtoull(argv[1], nullptr, 10) : 20;
    const unsigned threads = argc > 2 ? (unsigned)atoi(argv[2]) : 8;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
    std::vector<std::vector<Stats>> res(threads, std::vector<Stats>(8));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; t++) pool.emplace_back(worker, seed * 1000 + t, millions * 1000000ull, res[t].data());
    for (auto &t : pool) t.join();
    static const char *NAME[8] = {"plain", "dyadic", "short dirs", "long dirs", "Sun-like", "dyadic 2", "zero components", "hostile"};
    uint64_t bad = 0, n = 0;
    for (int c = 0; c < 8; c++) {
        Stats s;
        for (unsigned t = 0; t < threads; t++) { s.n += res[t][c].n; s.sure += res[t][c].sure; s.sure_hit += res[t][c].sure_hit; s.sure_in += res[t][c].sure_in; s.bad += res[t][c].bad; s.ref_hit += res[t][c].ref_hit; }
        printf("%-16s %12llu boxes  reference hits %5.1f %%  quick answer stands %6.2f %% (hits %5.1f %%, of them inside %5.1f %%)  mismatches %llu\n", NAME[c],
               (unsigned long long)s.n, 100.0 * s.ref_hit / (s.n ? s.n : 1), 100.0 * s.sure / (s.n ? s.n : 1), 100.0 * s.sure_hit / (s.sure ? s.sure : 1),
               100.0 * s.sure_in / (s.sure_hit ? s.sure_hit : 1), (unsigned long long)s.bad);
        bad += s.bad; n += s.n;
    }
    printf("total %llu boxes, %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad);
    return bad ? 1 : 0;
}
