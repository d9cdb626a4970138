// device_scene.h — device-side scene layout, primitive intersectors and BVH traversal.
//
// Device layout (built by the uploader from the reference's canonical quad array):
//   recs  : float4[4*R], one 64-byte record per INTERIOR node (pre-order; the record of an interior
//           lower child directly follows its parent's), holding the boxes of BOTH children so that one
//           fetch of one 64-byte line serves two box tests:
//             [4r]   = { lo.bbmin.xyz, bits(lo ref) }     ref = record index of an interior child, or
//             [4r+1] = { lo.bbmax.xyz, bits(hi ref) }           bit31 | first primitive index of a leaf child
//                                                               (| bit30 when the leaf is one or two triangles, | bit29 when two)
//             [4r+2] = { hi.bbmin.xyz, 0 }
//             [4r+3] = { hi.bbmax.xyz, 0 }
//           The root's own box and ref travel in the Scene struct.
//   prims : float4[3*P], fixed 48-byte records, primitive p at [3p..3p+2]:
//             sphere   { c.xyz, T }{ r, 0, 0, 0 }{ 0 }
//             disc     { c.xyz, T }{ n.xyz, r }{ 0 }
//             triangle { v0.xyz, T }{ e1.xyz, 0 }{ e2.xyz, 0 }     e1 = v1-v0, e2 = v2-v0 (fp32)
//             cone     { c1.xyz, T }{ axis.xyz, len }{ r1, widthCoeff, cosB, dotAxC1 }
//           T = bits(type | count << 2): the FIRST primitive of a leaf carries the leaf's primitive count
#pragma once
#include "device_math.h"

// The quick box answer (box_quick.h: certainly the reference's, or withdrawn) on the device's own instructions: v_min / v_max drop a
// NaN operand, v_med3 turns into min3 when it meets one, |x| is a source modifier, the fma is explicit.
#define GQ_FN __device__ __forceinline__
#define GQ_HOST_FN __host__ __device__ static inline
GQ_FN float gq_min(float a, float b) { return fminf(a, b); }
GQ_FN float gq_max(float a, float b) { return fmaxf(a, b); }
GQ_FN float gq_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
GQ_FN float gq_fma(float a, float b, float c) { return fmaf(a, b, c); }
GQ_FN float gq_abs(float a) { return fabsf(a); }
#include "box_quick.h"

namespace gd {

enum { P_SPHERE = 0, P_DISC = 1, P_TRIANGLE = 2, P_CONE = 3 };
#define GD_VISIBILITY_OFFSET 1.0e-4f
#define GD_NO_PRIM 0xffffffffu

#define GD_REF_LEAF 0x80000000u
#define GD_REF_TRIS 0x40000000u  ///< with GD_REF_LEAF: the leaf holds only triangles, one or two of them
#define GD_REF_TWO 0x20000000u   ///< with GD_REF_TRIS: two of them (both records are fetched at once; a single-triangle leaf fetches one)
#define GD_REF_SMALL 0x10000000u ///< with GD_REF_LEAF, without GD_REF_TRIS: one or two primitives of any type (GD_REF_TWO: two) — fetched at once like a triangle pair
#define GD_REF_INDEX 0x0fffffffu

// Visiting order of a closest-hit query. The reference always descends lower-then-upper (shaders/bvh_intersection.glsl:432-441),
// prunes a node whose box is entered beyond the closest hit so far (:398) and keeps the strictly closer hit (:416): among the hits with
// the smallest parameter the FIRST primitive in its depth-first order = the one with the lowest address wins.
//
// DEFAULT (round 5): every walk of the product keeps that order to the letter (the !NEAREST code below; the GD_REF_ORDER kernel variants).
// It is the only order PROVEN to return the reference's winner, and here is why no other one can be: the reference accepts whatever
// parameter its intersectors compute. Moeller-Trumbore at a grazing angle below ~1e-5 rad (shaders/triangle.glsl:50-76) divides two sums
// that have cancelled to a few ulps: the quotient is a small dyadic number — 0.8, 0.25, 0.03125 — unrelated to where the triangle is,
// and u and v, as arbitrary, sometimes pass. Such a PHANTOM hit can lie far in front of the triangle's own leaf box. The reference
// returns it iff its walk reaches that leaf while it holds nothing closer than the box's entry — a property of ITS order. Any walk that
// visits another child first and prunes with what it found there can skip that leaf, never test the triangle, and so cannot know: no
// certificate computed from the primitives a walk DID test covers a primitive it did not. tests/golden/order_adversary.npz holds 16
// such scenes (four primitives each: the phantom's triangle, a disc between phantom and box, two fillers), found in 2.5 s of search,
// expected values from the reference's own GLSL on llvmpipe; the reference-order walk returns them bit for bit, the walk below does not.
//
// OPT-IN (gpuart_hip_set_nearest_first / GPUART_HIP_NEAREST_MIN_PRIMS; the default of round 4): nearest-child-first WITH a certificate
// (NEAREST template arguments below) — ~17 % fewer node visits per query, 10 % less time per 1080p pass of cfg3 (0.74 against 0.82 ms),
// 17 % less for one pass alone. If every box were conservative for what it holds — entered no later than any hit inside it — the
// winner would not depend on the visiting order, and a walk that enters the nearer child first, stacks the other one and lets the lower
// primitive index win equal parameters would return it. In fp32 boxes are NOT always conservative, in three ways of growing size:
//   (1) rounding: a box entry parameter and a primitive's own hit parameter are computed two ways, and where the hit lies on the box's
//       face (flat axis-aligned triangles, a sphere's axis-extreme points, a triangle's extreme vertex) either may come out a few ulps
//       larger: 3 of 1e9 rays of cfg3, every tenth scene of coplanar lattice geometry (profiles/r04/nearest_child_first.txt);
//   (2) "odd" boxes: the face the ray enters through fails its own test by rounding at an edge, the reference's running minimum falls
//       on the face the ray leaves through, and the box claims to be entered beyond hits inside it by up to its whole depth (1 ray in ~1e8);
//   (3) phantom hits (above): off by any factor. NOT covered by what follows.
// The certificate covers (1) and (2):
//   * pruning is widened by GD_NEAREST_BAND: a node is skipped only if its entry parameter exceeds closest x BAND, so every
//     primitive within the band of the final hit is tested whatever the order;
//   * the query tracks the runner-up (second smallest accepted parameter) and whether the winner is "loose": some box on its
//     path is entered beyond the winner's own parameter (only then can the reference have pruned it) — its leaf's box, in fact:
//     without odd boxes entry parameters never decrease towards the leaves (aabb_entry);
//   * a box whose reported entry parameter is not its slab entry (aabb_entry's `odd`) marks the query;
//   * a finished query that met an odd box, or whose winner is adrift of its boxes by more than the band, or whose winner is loose
//     AND has a runner-up within the band, is walked again in the reference's order (about ten queries per million: 55 of 4.7e6 per
//     1080p pass of cfg3) — trav_settle;
//   * trees whose boxes do not bound their contents at all (hostile input: converter.h prim_in_box) never walk nearest-first.
// What that proves: let (t*, p*) be the nearest-first result, R the reference's. IF every primitive the reference accepts has a computed
// parameter no smaller than the entry parameter of each of its boxes / (1 + eps), with (1 + eps)^2 <= BAND — eps <= 0.195 % for
// BAND = 1 + 2^-8 —, then R was tested by the nearest-first walk, so t_R >= t*; the reference can only have missed p* through a box on
// p*'s path entered beyond t* (p* loose) while it held another hit with t* <= t < that entry (a runner-up within the band) — and when
// it did test p*, the winner by (parameter, index) is the same in both walks. The IF is (3)'s complement; it fails for a triangle
// grazed below ~1e-4 rad (relative error of the computed parameter ~ 8 u / |cos(ray, normal)|), a disc likewise, and no walk can check
// it for primitives it does not test. Measured: tools/order_soak.py, 3.7e11 rays of six scenes against the reference-order kernels
// without a differing pixel (profiles/r04/order_soak_final.txt) — soak-verified, not proven; hence opt-in. The counting "reference work"
// variants, trees with irregular boxes, Sun-shadow queries and the one-thread-per-pixel kernels keep the reference's order in any case.
#ifndef GD_NEAREST
#define GD_NEAREST 1
#endif
#ifndef GD_NEAREST_DIRECT
#define GD_NEAREST_DIRECT 0  ///< the persistent direct-lighting kernel too (measured slower: kernels_pipeline.h)
#endif
#ifndef GD_NEAREST_BAND
#define GD_NEAREST_BAND 1.00390625f  ///< 1 + 2^-8 (free up to 2^-8, +2.5 % time at 2^-6: profiles/r04/nearest_child_first.txt)
#endif
#ifndef GD_CERT_ODD
#define GD_CERT_ODD 1      // ablation switches of the certificate's parts (tools/ab_build.sh; never 0 in the product build)
#endif
#ifndef GD_CERT_HITS
#define GD_CERT_HITS 1
#endif
// Round 6, the step's bookkeeping (profiles/r06/step_isa_budget.txt); both bit-identical by construction, 0 = the code of round 5 for A/B:
#ifndef GD_POP_WHOLE
#define GD_POP_WHOLE 1     ///< TravStack::pop reads an entry's three words at once (one LDS round trip per accepted pop instead of two)
#endif
#ifndef GD_ADDR32
#define GD_ADDR32 1        ///< node records / triangle leaves addressed as base + 32-bit byte offset (converter.h bounds both arrays below 4 GB)
#endif
#ifndef GD_QUICK_BOXES
#define GD_QUICK_BOXES 1   ///< fast-form box tests try the quick answer of box_quick.h first (0: always the six face tests; A/B builds)
#endif
#define GD_PRIM_LOOSE 0x40000000u  ///< in Trav::hit_prim during a NEAREST walk: a box on the winner's path is entered beyond the winner's parameter
#define GD_PRIM_ADRIFT 0x20000000u ///< ... beyond the winner's parameter x BAND: the parameter and its boxes disagree by more than rounding
#define GD_PRIM_FLAGS (GD_PRIM_LOOSE | GD_PRIM_ADRIFT)

struct Scene {
    const float4 *__restrict__ recs;
    const float4 *__restrict__ prims;
    float root_min[3], root_max[3];
    uint32_t root_ref;    ///< record index, or GD_REF_LEAF | first primitive
    uint32_t exact_boxes; ///< some box of the tree is irregular (min > max or NaN on an axis): every box test takes the comparison form
    float box_slack;      ///< box_quick.h's slack constant for this tree (gq_slack_of_tree; +inf: every quick answer is withdrawn)
};


struct Ray {
    F3 o, d;
};

struct Surface {
    float pos;  ///< ray parameter; < 0: miss
    F3 p, n;    ///< intersection point, unit normal facing the ray origin
    int ptype;  ///< -1: miss
};

struct WorkCounters {
    uint32_t rays, nodes, prims[4];
    uint32_t steps, steps_top;  ///< interior-node visits (record fetches); those of records the uploader marked "top of the tree"
    uint32_t rewalks;           ///< queries a nearest-first walk could not certify and walked again in the reference's order (trav_settle)
};

// ---- reference shaders/sphere.glsl:31-70 -----------------------------------------------------
GD_FN void sphere_hit(const Ray &r, F3 center, float radius, float &pos, F3 &p, F3 &n) {
    F3 m = r.o - center;
    float a = dot3(r.d, r.d);
    float b = 2 * dot3(r.d, m);
    float c = dot3(m, m) - radius * radius;
    float delta = b * b - 4 * a * c;
    if (delta >= 0) {
        float sd = sqrtf(delta);
        float k1 = (-b + sd) / (a + a);
        float k2 = (-b - sd) / (a + a);
        if (k1 < GD_VISIBILITY_OFFSET) pos = k2;
        else if (k2 < GD_VISIBILITY_OFFSET) pos = k1;
        else pos = (k1 < k2 ? k1 : k2);
        p = r.o + r.d * pos;
        n = normalize3(p - center);
        if (dot3(r.o - p, n) < 0) n = -n;
    } else
        pos = -1;
}

// ---- reference shaders/disc.glsl:30-72 -------------------------------------------------------
GD_FN void disc_hit(const Ray &r, F3 center, float radius, F3 dn, float &pos, F3 &p, F3 &n) {
    pos = -1;
    float tmp = dot3(r.d, dn);
    if (fabsf(tmp) < 1.0e-8f) return;
    float k = dot3(dn, center - r.o) / tmp;
    if (k <= 0) return;
    F3 q = k * r.d + r.o;
    F3 d = q - center;
    if (dot3(d, d) <= radius * radius) {
        pos = k;
        p = q;
        n = (dot3(r.o - center, dn) > 0) ? dn : -dn;
    }
}

// ---- reference shaders/triangle.glsl:33-82 (edges precomputed at upload, same fp32 subtraction) ---
GD_FN void triangle_hit(const Ray &r, F3 v0, F3 edge1, F3 edge2, float &pos, F3 &p, F3 &n) {
    pos = -1;
    F3 pvec = cross3(r.d, edge2);
    float det = dot3(edge1, pvec);
    if (fabsf(det) < 1.0e-10f) return;
    float invDet = 1 / det;
    F3 tvec = r.o - v0;
    float du = dot3(tvec, pvec);
    float u = du * invDet;
    if (u < 0 || u > 1) return;
    F3 qvec = cross3(tvec, edge1);
    float dv = dot3(r.d, qvec);
    float v = dv * invDet;
    // `u + v > 1` is evaluated by llvmpipe as (du + dv) * invDet (NIR distributes the common factor)
    if (v < 0 || (du + dv) * invDet > 1) return;
    pos = dot3(edge2, qvec) * invDet;
    p = r.o + r.d * pos;
    n = normalize3(cross3(edge1, edge2));
    if (dot3(r.o - p, n) < 0) n = -n;
}

// ---- reference shaders/cone.glsl:30-135 ------------------------------------------------------
GD_FN void cone_hit(const Ray &r, F3 c1, float r1, F3 ax, float axLen, float widthCoeff, float cosB, float dotAxC1,
                    float &pos, F3 &p, F3 &n) {
    const float CONE_TOLERANCE = 1.0e-7f;
    pos = -1;
    float axd = dot3(ax, r.d), axs = dot3(ax, r.o);
    F3 D = axd * ax;
    F3 E = -r.d;
    F3 F = ((c1 + axs * ax) - dotAxC1 * ax) - r.o;
    float G = widthCoeff * axd;
    float H = (widthCoeff * axs + r1) - widthCoeff * dotAxC1;  // llvmpipe's evaluation order
    float A = ((dot3(D, D) + dot3(E, E)) + 2 * dot3(D, E)) - G * G;
    float B = 2 * dot3(F, D + E) - 2 * G * H;
    float C = dot3(F, F) - H * H;
    if (fabsf(A) < CONE_TOLERANCE) return;
    float delta = B * B - 4 * A * C;
    if (delta < CONE_TOLERANCE) return;
    float sq = sqrtf(delta);
    float k1 = (-B + sq) / (A + A);
    float k2 = (-B - sq) / (A + A);
    F3 p1 = r.o + k1 * r.d, p2 = r.o + k2 * r.d;
    float t1 = dot3(ax, p1 - c1), t2 = dot3(ax, p2 - c1);
    bool on1 = t1 >= 0 && t1 <= axLen;
    bool on2 = t2 >= 0 && t2 <= axLen;
    if (k1 < GD_VISIBILITY_OFFSET && on2) { pos = k2; p = p2; }
    else if (k2 < GD_VISIBILITY_OFFSET && on1) { pos = k1; p = p1; }
    else if ((k1 < k2 && on1 && on2) || (on1 && !on2)) { pos = k1; p = p1; }
    else if ((k2 < k1 && on1 && on2) || (!on1 && on2)) { pos = k2; p = p2; }
    else return;
    if (pos > 0) {
        F3 proj = c1 + dot3(ax, p - c1) * ax;
        F3 n1 = normalize3(p - proj);
        float u = cosB - dot3(n1, ax);
        n = normalize3(u * ax + n1);
        if (dot3(n, r.d) > 0) n = -n;
    }
}

/// Ray parameter of one triangle record (the arithmetic of triangle.glsl:50-76, including the (du+dv)*invDet form and
/// dot3's order). Straight-line: every rejection is a term of one predicate; a rejected triangle yields -1, and so does
/// one closer than VISIBILITY_OFFSET (bvh_intersection.glsl:216-217). A version that carried the two triangles of a leaf
/// in packed 2-wide vectors (v_pk_mul/add_f32) was 2 % slower than two of these (tools/ab.py).
GD_FN float triangle_t(float rox, float roy, float roz, float rdx, float rdy, float rdz, float4 a0, float4 a1, float4 a2) {
    const float px = rdy * a2.z - rdz * a2.y, py = rdz * a2.x - rdx * a2.z, pz = rdx * a2.y - rdy * a2.x;
    const float det = (a1.z * pz + a1.y * py) + a1.x * px;
    const float inv = 1 / det;
    const float tx = rox - a0.x, ty = roy - a0.y, tz = roz - a0.z;
    const float du = (tz * pz + ty * py) + tx * px;
    const float u = du * inv;
    const float qx = ty * a1.z - tz * a1.y, qy = tz * a1.x - tx * a1.z, qz = tx * a1.y - ty * a1.x;
    const float dv = (rdz * qz + rdy * qy) + rdx * qx;
    const float v = dv * inv;
    const float w = (du + dv) * inv;
    const float t = ((a2.z * qz + a2.y * qy) + a2.x * qx) * inv;
    bool ok = !(fabsf(det) < 1.0e-10f) & !(u < 0) & !(u > 1) & !(v < 0) & !(w > 1) & !(t < GD_VISIBILITY_OFFSET);
    return ok ? t : -1.0f;
}

/// One primitive record against a ray (reference CheckBVHPrimitiveIntersection,
/// shaders/bvh_intersection.glsl:125-223, including its `pos < VISIBILITY_OFFSET -> -1` cut).
/// TYPES: bit t set = primitives of type t may occur (the uploader knows which types a scene holds; code for absent
/// types is not generated, which is worth 15 VGPRs — a fifth wave per SIMD — in the BVH-query kernel).
#define GD_ALL_TYPES 0xF
#define GD_EXACT_BOXES 0x10  ///< beside a type mask: the kernel variant for trees with irregular boxes (box tests in comparison form)
#define GD_REF_ORDER 0x20    ///< beside a type mask: regular boxes, every walk keeps the reference's order — the product's default since round 5
                             ///< (top of this file); without it: the opt-in nearest-child-first variants
#define GD_NEAREST_OF(TYPES) (GD_NEAREST && !((TYPES) & (GD_EXACT_BOXES | GD_REF_ORDER)))
/// How a kernel tests boxes: the fast (med3) form, the exact comparison form, or whichever Scene::exact_boxes asks for
/// (kernels off the fast path: one wave-uniform branch per step).
enum { GD_BOXES_FAST = 0, GD_BOXES_EXACT = 1, GD_BOXES_RUNTIME = 2 };
#define GD_BOXES_OF(TYPES) (((TYPES) & GD_EXACT_BOXES) ? GD_BOXES_EXACT : GD_BOXES_FAST)
template <int TYPES = GD_ALL_TYPES>
GD_FN void prim_hit(const Ray &r, float4 q0, float4 q1, float4 q2, float &pos, F3 &p, F3 &n, int &ptype) {
    ptype = (int)(__float_as_uint(q0.w) & 3u);
    pos = -1; p = f3(0, 0, 0); n = f3(0, 0, 0);
    if ((TYPES >> P_TRIANGLE & 1) && ptype == P_TRIANGLE) triangle_hit(r, xyz(q0), xyz(q1), xyz(q2), pos, p, n);
    else if ((TYPES >> P_SPHERE & 1) && ptype == P_SPHERE) sphere_hit(r, xyz(q0), q1.x, pos, p, n);
    else if ((TYPES >> P_DISC & 1) && ptype == P_DISC) disc_hit(r, xyz(q0), q1.w, xyz(q1), pos, p, n);
    else if ((TYPES >> P_CONE & 1) && ptype == P_CONE) cone_hit(r, xyz(q0), q2.x, xyz(q1), q1.w, q2.y, q2.z, q2.w, pos, p, n);
    if (pos < GD_VISIBILITY_OFFSET) pos = -1;
}

/// Ray/AABB entry test (reference IntersectsAABB, shaders/bvh_intersection.glsl:229-354) without branches
/// and almost without scalar mask arithmetic: each of the six faces yields a candidate parameter that is
/// either its plane parameter k (when k >= 0 and the hit point lies within the face, bounds inclusive) or
/// +inf; the result is the minimum candidate, exactly the running minimum of the reference.
///   * `v within [lo,hi]` is evaluated as med3(v, lo, hi) == v, which equals (v >= lo && v <= hi) for every
///     box with lo <= hi; +-0 and NaN in v behave like the two comparisons. It does NOT for an irregular box
///     (lo > hi or NaN on an axis: a sphere or disc with a negative, infinite or NaN radius, a NaN coordinate):
///     the reference can still hit such a box through the two planes of its one irregular axis — those faces
///     only check the other two axes. Trees that hold an irregular box (the uploader sets Scene::exact_boxes) run
///     every box test in the EXACT form, two comparisons per bound pair, behind a wave-uniform branch; all others
///     never pay for it.
///   * the reference's `rdir.c != 0` guards need no code: with rdiv.c = +-inf the plane parameter is +-inf or
///     NaN, and then either `k >= 0` fails or the hit point is +-inf/NaN and fails the face check.
/// Returns false on a miss; pos = -1 when the origin is inside (inclusive).
GD_FN bool within(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi) == v; }

GD_FN float face_candidate(float k, float a0, float a1, float lo_a, float hi_a, float b0, float b1, float lo_b, float hi_b) {
    const float INF = __builtin_inff();
    const float a = a0 + k * a1, b = b0 + k * b1;
    float c = (k >= 0) ? k : INF;
    c = within(a, lo_a, hi_a) ? c : INF;
    c = within(b, lo_b, hi_b) ? c : INF;
    return c;
}

/// The comparison form: IntersectsAABB statement by statement (shaders/bvh_intersection.glsl:229-354), for trees with
/// irregular boxes — the `rdir.c != 0` guards are real here (a plane of an infinite box reached with k = +inf is a
/// candidate the guard must be able to veto), a candidate with k >= 1e19 counts as an intersection but leaves pos at
/// 1e19, and the running minimum keeps the FIRST of +0 / -0 as `if (k < pos)` does.
GD_FN bool aabb_entry_exact(const Ray &r, F3 rdiv, F3 bmin, F3 bmax, float &pos) {
    const bool inside = (r.o.x >= bmin.x) & (r.o.y >= bmin.y) & (r.o.z >= bmin.z) & (r.o.x <= bmax.x) & (r.o.y <= bmax.y) & (r.o.z <= bmax.z);
    bool hit = false;
    float p = 1.0e+19f;
#define GD_FACE(dc, plane, oc, rdivc, a0, a1, lo_a, hi_a, b0, b1, lo_b, hi_b)                                      \
    {                                                                                                              \
        const float k = ((plane) - (oc)) * (rdivc);                                                                \
        const float a = (a0) + k * (a1), b = (b0) + k * (b1);                                                      \
        const bool ok = ((dc) != 0) & (k >= 0) & (a >= (lo_a)) & (a <= (hi_a)) & (b >= (lo_b)) & (b <= (hi_b));    \
        hit |= ok;                                                                                                 \
        p = (ok & (k < p)) ? k : p;                                                                                \
    }
    GD_FACE(r.d.x, bmin.x, r.o.x, rdiv.x, r.o.y, r.d.y, bmin.y, bmax.y, r.o.z, r.d.z, bmin.z, bmax.z)
    GD_FACE(r.d.x, bmax.x, r.o.x, rdiv.x, r.o.y, r.d.y, bmin.y, bmax.y, r.o.z, r.d.z, bmin.z, bmax.z)
    GD_FACE(r.d.y, bmin.y, r.o.y, rdiv.y, r.o.x, r.d.x, bmin.x, bmax.x, r.o.z, r.d.z, bmin.z, bmax.z)
    GD_FACE(r.d.y, bmax.y, r.o.y, rdiv.y, r.o.x, r.d.x, bmin.x, bmax.x, r.o.z, r.d.z, bmin.z, bmax.z)
    GD_FACE(r.d.z, bmin.z, r.o.z, rdiv.z, r.o.x, r.d.x, bmin.x, bmax.x, r.o.y, r.d.y, bmin.y, bmax.y)
    GD_FACE(r.d.z, bmax.z, r.o.z, rdiv.z, r.o.x, r.d.x, bmin.x, bmax.x, r.o.y, r.d.y, bmin.y, bmax.y)
#undef GD_FACE
    pos = inside ? -1.0f : p;
    return inside | hit;
}

/// `odd` (fast form, when asked for): the box is hit from outside, but the entry parameter the reference's running minimum
/// arrives at is NOT the slab entry — the largest of the three axes' nearer plane parameters, which is what it equals, bit
/// for bit, whenever the face the ray enters through passes its own test and no face passes early. Boxes that are not odd
/// therefore report their slab entry, and slab entries never decrease from a box to a box nested in it (the uploader checks the
/// nesting: converter.h): fp32 subtraction, multiplication by the ray's 1/d, min and max are all monotone. That is what lets a
/// walk take the entry parameter of a leaf for the largest one on the path to it (take_hit's `loose`). Where rounding fails that face (a ray through an
/// edge or a corner) the minimum falls on the face the ray LEAVES through: the box then reports an entry parameter larger than
/// hits inside it by up to its whole depth, and what the reference finds in it depends on when its walk gets there
/// (1 ray in ~1e8; device_scene.h top, trav_settle).
template <bool EXACT = false, bool WITH_ODD = false>
GD_FN bool aabb_entry(const Ray &r, F3 rdiv, F3 bmin, F3 bmax, float &pos, bool *odd = nullptr) {
    if (EXACT) return aabb_entry_exact(r, rdiv, bmin, bmax, pos);
    const bool inside = within(r.o.x, bmin.x, bmax.x) & within(r.o.y, bmin.y, bmax.y) & within(r.o.z, bmin.z, bmax.z);
    const float k0 = (bmin.x - r.o.x) * rdiv.x, k1 = (bmax.x - r.o.x) * rdiv.x;
    const float k2 = (bmin.y - r.o.y) * rdiv.y, k3 = (bmax.y - r.o.y) * rdiv.y;
    const float k4 = (bmin.z - r.o.z) * rdiv.z, k5 = (bmax.z - r.o.z) * rdiv.z;
    const float c0 = face_candidate(k0, r.o.y, r.d.y, bmin.y, bmax.y, r.o.z, r.d.z, bmin.z, bmax.z);
    const float c1 = face_candidate(k1, r.o.y, r.d.y, bmin.y, bmax.y, r.o.z, r.d.z, bmin.z, bmax.z);
    const float c2 = face_candidate(k2, r.o.x, r.d.x, bmin.x, bmax.x, r.o.z, r.d.z, bmin.z, bmax.z);
    const float c3 = face_candidate(k3, r.o.x, r.d.x, bmin.x, bmax.x, r.o.z, r.d.z, bmin.z, bmax.z);
    const float c4 = face_candidate(k4, r.o.x, r.d.x, bmin.x, bmax.x, r.o.y, r.d.y, bmin.y, bmax.y);
    const float c5 = face_candidate(k5, r.o.x, r.d.x, bmin.x, bmax.x, r.o.y, r.d.y, bmin.y, bmax.y);
    const float best = fminf(fminf(fminf(c0, c1), fminf(c2, c3)), fminf(c4, c5));
    // the reference starts its running minimum at 1e19 and only lowers it with `k < pos`: a face whose k is
    // >= 1e19 still counts as an intersection but leaves pos at 1e19
    const bool hit = best < __builtin_inff();
    pos = inside ? -1.0f : fminf(best, 1.0e+19f);
    if (WITH_ODD) {
        const float slab = fmaxf(fmaxf(fminf(k0, k1), fminf(k2, k3)), fminf(k4, k5));
        *odd = !inside & hit & (best != slab);  // (NaN parameters — a ray component of exactly 0 on a box plane — compare unequal: odd)
    }
    return inside | hit;
}

/// The quick answer for one box (box_quick.h) on aabb_entry's own plane parameters — the inside answer included: for the rays and
/// trees the slack vets, the signs of the parameters decide it as the reference's comparisons do; `cs` = gq_ray_slack of the ray. True:
/// `hit` and `pos` are what aabb_entry returns (and the box is not odd). False: withdrawn — ask aabb_entry.
GD_FN bool box_quick(const Ray &r, F3 rdiv, F3 bmin, F3 bmax, float cs, float &pos, bool &hit) {
    const float k0 = (bmin.x - r.o.x) * rdiv.x, k1 = (bmax.x - r.o.x) * rdiv.x;
    const float k2 = (bmin.y - r.o.y) * rdiv.y, k3 = (bmax.y - r.o.y) * rdiv.y;
    const float k4 = (bmin.z - r.o.z) * rdiv.z, k5 = (bmax.z - r.o.z) * rdiv.z;
    return gq_box(k0, k1, k2, k3, k4, k5, fabsf(rdiv.x), fabsf(rdiv.y), fabsf(rdiv.z), cs, pos, hit);
}

#ifdef GD_QUICK_CHECK
// Diagnostic build (never the product): every fast-form box test of a step is ALSO run through the six face tests and the quick
// answers that stand are compared with them. [0] boxes, [1] quick answers that stand, [2] steps, [3] steps that had to run the face
// tests (a lane withdrew), [4] standing answers that differ from the face tests (must stay 0), [5] of them: marked odd by the face tests.
__device__ unsigned long long g_quick_stats[8];
#endif

/// Running result of one query (the traversal state proper follows below).
struct Trav {
    float closest;
    uint32_t hit_prim;   ///< (| GD_PRIM_LOOSE during a NEAREST walk; trav_settle removes it)
    uint32_t node;       ///< DESCEND: record to fetch; LEAF: first primitive index
    float entry;         ///< box-entry parameter of the node entered last (on a path without odd boxes also the largest one on the path: aabb_entry)
    int state;
    float second;        ///< NEAREST walks: parameter of the runner-up (1e19: none)
};

/// A tested primitive's parameter into the running result; true if it is the closest hit now. The reference's walk meets the
/// primitives in address order and keeps the strictly closer one (shaders/bvh_intersection.glsl:416). A NEAREST walk gets the same
/// winner in any order by letting the lower index win equal parameters (signed compare: GD_NO_PRIM = -1 never loses a tie it
/// cannot have — closest starts at 1e19 and a hit AT 1e19 is not accepted by the reference's `<` either), and keeps what
/// trav_settle needs: the runner-up's parameter and whether the winner is loose (t.entry, the entry parameter of this leaf's box — the largest on the path
/// to this leaf, exceeds its parameter).
template <bool NEAREST>
GD_FN bool take_hit(Trav &t, float pos, uint32_t pi) {
    if (!(pos > 0)) return false;
    if (!NEAREST) {
        if (!(pos < t.closest)) return false;
        t.closest = pos;
        t.hit_prim = pi;
        return true;
    }
    const bool win = (pos < t.closest) | ((pos == t.closest) & ((int)pi < (int)(t.hit_prim & ~GD_PRIM_FLAGS)));
    if (!GD_CERT_HITS) {
        t.hit_prim = win ? pi : t.hit_prim;
        t.closest = win ? pos : t.closest;
        return win;
    }
    t.second = fminf(t.second, win ? t.closest : pos);  // (second >= closest unless it is the -inf of an odd box, which sticks)
    t.hit_prim = win ? (pi | (t.entry > pos ? GD_PRIM_LOOSE : 0u) | (t.entry > pos * GD_NEAREST_BAND ? GD_PRIM_ADRIFT : 0u)) : t.hit_prim;
    t.closest = win ? pos : t.closest;
    return win;
}

/// Tests the primitives of the leaf starting at primitive `first` (its count sits in the first record's
/// type word); keeps the strictly closer hit (the first one wins ties, reference
/// shaders/bvh_intersection.glsl:405-423). Returns true if ANY_HIT and something was hit.
template <bool ANY_HIT, bool COUNT, int TYPES = GD_ALL_TYPES, bool NEAREST = false>
GD_FN bool leaf_test(const Scene &sc, const Ray &r, uint32_t first, Trav &t, WorkCounters *wc) {
    uint32_t count = 1;
    for (uint32_t i = 0; i < count; i++) {
        uint32_t pi = first + i;
#if GD_ADDR32
        const float4 *pp = (const float4 *)((const char *)sc.prims + pi * 48u);
        float4 q0 = pp[0], q1 = pp[1], q2 = pp[2];
#else
        float4 q0 = sc.prims[3 * (size_t)pi], q1 = sc.prims[3 * (size_t)pi + 1], q2 = sc.prims[3 * (size_t)pi + 2];
#endif
        if (i == 0) {
            count = __float_as_uint(q0.w) >> 2;
            if (count == 0) break;  // empty leaf (empty scene)
        }
        float pos; F3 p, n; int ptype;
        prim_hit<TYPES>(r, q0, q1, q2, pos, p, n, ptype);
        if (COUNT) wc->prims[ptype & 3]++;
        if (take_hit<NEAREST>(t, pos, pi) && ANY_HIT) return true;
    }
    return false;
}

/// Leaf that holds one or two triangles (the common case of triangle meshes: the reference builds leaves of at
/// most two primitives). Which of the two it is travels in the leaf's ref, so both records of a pair are requested at
/// once and a single-triangle leaf requests only its own (every 16-byte request counts: the BVH queries are bound by the
/// vector-memory request pipeline). The first one wins ties (`pos < closest` is strict), as in the reference's loop.
template <bool ANY_HIT, bool COUNT, bool NEAREST = false>
GD_FN bool leaf_test_tris(const Scene &sc, const Ray &r, uint32_t first, bool two, Trav &t, WorkCounters *wc) {
#if GD_ADDR32
    const float4 *pa = (const float4 *)((const char *)sc.prims + first * 48u);
#else
    const float4 *pa = sc.prims + 3 * (size_t)first;
#endif
    const float4 a0 = pa[0], a1 = pa[1], a2 = pa[2];
    if (COUNT) wc->prims[P_TRIANGLE] += two ? 2 : 1;
    float tb = -1.0f;
    if (two) {
        const float4 b0 = pa[3], b1 = pa[4], b2 = pa[5];
        tb = triangle_t(r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, b0, b1, b2);
#ifdef GD_LEAF_PROBE  // measurement hook (never in the product build): one more triangle test per pair, result unused
        {
            float ox = r.o.x;
            asm volatile("" : "+v"(ox));
            const float tc = triangle_t(ox, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, b0, b1, b2);
            asm volatile("" ::"v"(tc));
        }
#endif
    }
    const float ta = triangle_t(r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, a0, a1, a2);
    if (take_hit<NEAREST>(t, ta, first) && ANY_HIT) return true;
    if (take_hit<NEAREST>(t, tb, first + 1) && ANY_HIT) return true;
    return false;
}

/// Leaf that holds one or two primitives of any type (what the reference's builder makes of everything but degenerate
/// input): the leaf's ref says which, so a pair's records are requested together instead of one memory round trip per
/// primitive, and nothing waits for the count in the first record. Same tests in the same order as the loop of `leaf_test`.
template <bool ANY_HIT, bool COUNT, int TYPES = GD_ALL_TYPES, bool NEAREST = false>
GD_FN bool leaf_test_small(const Scene &sc, const Ray &r, uint32_t first, bool two, Trav &t, WorkCounters *wc) {
#if GD_ADDR32
    const float4 *pa = (const float4 *)((const char *)sc.prims + first * 48u);
#else
    const float4 *pa = sc.prims + 3 * (size_t)first;
#endif
    const float4 a0 = pa[0], a1 = pa[1], a2 = pa[2];
    float tb = -1.0f;
    F3 p, n; int ptype;
    if (two) {
        const float4 b0 = pa[3], b1 = pa[4], b2 = pa[5];
        prim_hit<TYPES>(r, b0, b1, b2, tb, p, n, ptype);
        if (COUNT) wc->prims[ptype & 3]++;
    }
    float ta;
    prim_hit<TYPES>(r, a0, a1, a2, ta, p, n, ptype);
    if (COUNT) wc->prims[ptype & 3]++;
    if (take_hit<NEAREST>(t, ta, first) && ANY_HIT) return true;
    if (take_hit<NEAREST>(t, tb, first + 1) && ANY_HIT) return true;
    return false;
}

// ---- traversal stack: short per-lane ring in LDS + spill to global memory -------------------------
// Entry = (upper child's ref, parent's box-entry parameter, the child's own box-entry parameter).
// Entries [base, sp) live in the LDS ring (slot = index % RING, one column per
// lane -> conflict-free accesses); entries [0, base) live in a per-lane global spill column. The oldest
// entries (closest to the root, popped last) are the ones spilled, so spills are rare; any tree depth
// up to the reference's 1024 levels works without a separate code path.
#ifndef GD_RING
#define GD_RING 8
#endif

struct StackEntry {
    uint32_t ref;   ///< upper child: record index, or GD_REF_LEAF | first primitive
    float pe, he;   ///< box-entry parameter of the parent / of the child itself (GD_ENTRY_MISS: box not hit)
};

#if (GD_RING & (GD_RING - 1)) == 0
#define GD_RING_SLOT(i) ((i) & (GD_RING - 1))
#else
#define GD_RING_SLOT(i) ((i) % GD_RING)  // i < 1024 + GD_RING: a multiply-shift
#endif

#ifdef GD_RUN_TIMELINE
__device__ unsigned long long g_stack_events[4];  // diagnostic build: pushes, pushes that spill an entry to global memory, pops, pops that reload one
#define GD_STACK_EVENT(k) atomicAdd(&g_stack_events[k], 1ull)
#else
#define GD_STACK_EVENT(k)
#endif
struct TravStack {
    uint2 *ring_a;           ///< LDS, [GD_RING][BLOCK]: (ref, pe)
    float *ring_b;           ///< LDS, [GD_RING][BLOCK]: he
    uint4 *spill;            ///< global, [levels][total_lanes]; this lane's column is spill[level * spill_stride]
    uint32_t ring_stride;    ///< BLOCK
    uint32_t spill_stride;   ///< total lanes of the launch
    uint32_t sp, base;
    GD_FN void reset() { sp = 0; base = 0; }
    /// The replicas of a ray (thin-wave modes) read what its first replica stored: the stores of `if (writer)` must stay ahead
    /// of the other lanes' loads in the instruction stream. Per thread the compiler could legally turn "conditional store, then
    /// load" into "writer keeps the value, the others load" and run the others first; a wavefront-scope fence (no instruction:
    /// a wave executes its memory operations in order) plus a wave barrier pin the order.
    static GD_FN void replica_fence() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    /// `writer`: this lane performs the stores. A ray that several lanes carry as identical replicas (the thin-wave modes
    /// below) has ONE column; every replica keeps sp / base and reads the column, only the first one writes it.
    GD_FN void push(StackEntry e, bool writer = true) {
        if (writer) GD_STACK_EVENT(0);
        if (sp - base == GD_RING) {
            if (writer) GD_STACK_EVENT(1);
            uint32_t o = GD_RING_SLOT(base) * ring_stride;
            uint2 a = ring_a[o];
            if (writer) spill[(size_t)base * spill_stride] = make_uint4(a.x, a.y, __float_as_uint(ring_b[o]), 0);
            replica_fence();
            base++;
        }
        uint32_t o = GD_RING_SLOT(sp) * ring_stride;
        if (writer) {
            ring_a[o] = make_uint2(e.ref, __float_as_uint(e.pe));
            ring_b[o] = e.he;
        }
        replica_fence();
        sp++;
    }
    GD_FN StackEntry pop(bool writer = true) {  // precondition: sp > 0
        if (writer) GD_STACK_EVENT(2);
        if (sp == base) {
            if (writer) GD_STACK_EVENT(3);
            base--;
            uint4 v = spill[(size_t)base * spill_stride];
            uint32_t o = GD_RING_SLOT(base) * ring_stride;
            if (writer) {
                ring_a[o] = make_uint2(v.x, v.y);
                ring_b[o] = __uint_as_float(v.z);
            }
            replica_fence();
        }
        sp--;
        uint32_t o = GD_RING_SLOT(sp) * ring_stride;
        uint2 a = ring_a[o];
        StackEntry e;
        e.ref = a.x; e.pe = __uint_as_float(a.y); e.he = ring_b[o];
#if GD_POP_WHOLE
        // Keep the entry's three words in the two LDS reads issued here: left alone the compiler reads the parameters first (4 + 4 bytes),
        // compares, and fetches the ref with a THIRD read once the entry is accepted — a second LDS round trip in the dependent chain
        // pop -> enter -> record address -> fetch of every step that ends in a pop.
        asm volatile("" : "+v"(e.ref), "+v"(e.pe), "+v"(e.he));
#endif
        return e;
    }
};

/// Traversal state of one ray. In the reference's order (every walk that is not an ordered NEAREST one: the counting and
/// one-thread-per-pixel kernels, trees with irregular boxes, Sun-shadow queries, a query that trav_settle sends round again)
/// the walk visits exactly the nodes, in exactly the order, of the
/// reference's stackless parent-pointer walk (shaders/bvh_intersection.glsl:360-457): lower child
/// first, prune on `entry > closest`. Where the reference re-tests a parent's box when it returns from
/// the lower child, the parent's entry parameter kept on the stack is compared with the current closest
/// hit (same value: the test is a pure function of node and ray). The upper child's box is tested when
/// its parent's record is fetched (both boxes share one 64-byte record) and the result waits on the
/// stack; whether it is *used* is decided exactly where the reference decides it, at pop time.
// odd: waiting at a leaf; a leaf's state is 1 | its ref's GD_REF_TRIS / GD_REF_TWO / GD_REF_SMALL bits moved down to 8 / 4 / 2
enum { TRAV_DESCEND = 0, TRAV_LEAF = 1, TRAV_DONE = 2, TRAV_LEAF_ONE = 3, TRAV_LEAF_PAIR = 7, TRAV_LEAF_TRI1 = 9, TRAV_LEAF_TRIS = 13 };
static_assert((GD_REF_TRIS >> 27) == 8 && (GD_REF_TWO >> 27) == 4 && (GD_REF_SMALL >> 27) == 2, "leaf states are derived from the ref's flag bits");

#define GD_ENTRY_MISS 3.0e+38f  // stack marker: the upper child's box is not hit at all


/// Enters a child whose box test passed: a leaf waits for its primitive tests, an interior child for its record.
GD_FN void trav_enter(Trav &t, uint32_t ref, float entry) {
    t.entry = entry;  // (a leaf's: take_hit compares the accepted hit with it)
    if (ref & GD_REF_LEAF) {
        t.node = ref & GD_REF_INDEX;
        t.state = (int)(1u | ((ref >> 27) & 14u));  // TRAV_LEAF, _ONE, _PAIR, _TRI1 or _TRIS (a leaf with more than two primitives keeps all three bits clear)
    } else {
        t.node = ref;
        t.state = TRAV_DESCEND;
    }
}

#ifdef GD_STEP_STATS
__device__ unsigned long long g_pop_stats[4];  // [0] trips of trav_pop's loop as waves execute them, [1] lanes in those trips, [2] calls (wave level), [3] lanes in the calls
#endif
/// Returns to the nearest pending upper child that is still worth visiting (or finishes).
/// `band`: 1 in the reference's order; GD_NEAREST_BAND in an ordered NEAREST walk.
template <bool COUNT>
GD_FN void trav_pop(Trav &t, TravStack &st, WorkCounters *wc, bool writer = true, float band = 1.0f) {
    const float limit = t.closest * band;  // (x 1.0f is exact)
#ifdef GD_STEP_STATS
    {
        const unsigned long long act = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) { atomicAdd(&g_pop_stats[2], 1ull); atomicAdd(&g_pop_stats[3], (unsigned long long)__popcll(act)); }
    }
#endif
    for (;;) {
#ifdef GD_STEP_STATS
        {   // diagnostic build: trips of this loop as the WAVE executes them (one count per trip with any lane in it) and lane-level pops
            const unsigned long long act = __ballot(1);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) { atomicAdd(&g_pop_stats[0], 1ull); atomicAdd(&g_pop_stats[1], (unsigned long long)__popcll(act)); }
        }
#endif
        if (st.sp == 0) { t.state = TRAV_DONE; return; }
        StackEntry e = st.pop(writer);
        if (e.pe > limit) continue;       // the reference's parent re-test fails: skip the upper child
        if (COUNT) wc->nodes++;           // the reference tests the upper child's box now
        if (e.he > limit) continue;       // box missed (GD_ENTRY_MISS), or entered beyond the closest hit
        trav_enter(t, e.ref, e.he);
        return;
    }
}

/// The band of a lane's walk: GD_NEAREST_BAND if it visits the nearer child first (`ordered`), 1 in the reference's order.
GD_FN float trav_band(bool ordered) { return ordered ? GD_NEAREST_BAND : 1.0f; }

/// A NEAREST walk has just finished (state DONE): removes the bookkeeping bit from the result and says whether the query must
/// be walked again in the reference's order (trav_init, then steps with band 1): its winner is loose and a runner-up lies
/// within the band — the one constellation in which the reference's order can have made another primitive win by rounding —,
/// or the winner is adrift of its boxes by more than the band (a primitive whose computed parameter is far off: the reference
/// may have pruned it whatever else it held), or the walk met an odd box (aabb_entry: second = -inf), whose entry parameter
/// says nothing about what is inside.
template <bool NEAREST>
GD_FN bool trav_settle(Trav &t, bool ordered) {
    if (!NEAREST) return false;
    const bool hit = (int)t.hit_prim >= 0;
    const bool again = ordered & ((t.second < 0) | (hit & (((t.hit_prim & GD_PRIM_ADRIFT) != 0) |
                                                          (((t.hit_prim & GD_PRIM_LOOSE) != 0) & (t.second <= t.closest * GD_NEAREST_BAND)))));
    t.hit_prim = hit ? (t.hit_prim & ~GD_PRIM_FLAGS) : t.hit_prim;
    return again;
}

/// `certify`: the walk that follows is an ordered NEAREST one — an odd root box (aabb_entry) marks the query like any other odd box.
template <int BOXES = GD_BOXES_RUNTIME>
GD_FN void trav_init(const Scene &sc, const Ray &r, F3 rdiv, Trav &t, TravStack &st, WorkCounters *wc, bool count, bool certify = false) {
    t.closest = 1e+19f;
    t.second = 1e+19f;
    t.hit_prim = GD_NO_PRIM;
    t.node = 0;
    t.entry = 0;
    st.reset();
    if (count) { wc->rays++; wc->nodes++; }
    // Rays that start inside the root box (nearly all: bounce and shadow rays leave surfaces, the camera usually stands
    // inside the scene's box) are "hit, entry -1" by the inclusive inside test alone; the six face tests are only run
    // when some lane of the wave starts outside.
    const F3 bmin = f3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), bmax = f3(sc.root_max[0], sc.root_max[1], sc.root_max[2]);
    float entry = -1.0f;
    bool hit = true;
    if (BOXES == GD_BOXES_EXACT || (BOXES == GD_BOXES_RUNTIME && sc.exact_boxes)) {
        hit = aabb_entry<true>(r, rdiv, bmin, bmax, entry);
    } else {
        const bool inside = within(r.o.x, bmin.x, bmax.x) & within(r.o.y, bmin.y, bmax.y) & within(r.o.z, bmin.z, bmax.z);
        if (__ballot(!inside) != 0) {
            bool odd = false;
            hit = aabb_entry<false, true>(r, rdiv, bmin, bmax, entry, &odd);
            if (certify & odd) t.second = -__builtin_inff();
        }
    }
    if (hit) trav_enter(t, sc.root_ref, entry);  // entry > 1e19 cannot happen
    else t.state = TRAV_DONE;
#ifdef GD_ROOT_PROBE
    // Measurement hook (never in the product build): what the fetch of the root's record costs a ray — every lane that starts a
    // query fetches record 0 once more (same address in all lanes, as the real fetch of that record is). Passing the record in
    // kernel arguments instead could at best save this much; built for real (the visit of the root fused into this function,
    // record from scalar loads) it LOST 1.6-3.5 %: the fused visit is issued for the few lanes that start a query, beside the
    // wave's regular box step (profiles/r03/request_trims.txt).
    for (int k = 0; k < GD_ROOT_PROBE; k++) {
        const float4 *rp = sc.recs;
        asm volatile("" : "+v"(rp));
        float4 d0 = rp[0], d1 = rp[1], d2 = rp[2], d3 = rp[3];
        asm volatile("" ::"v"(d0.x), "v"(d0.w), "v"(d1.x), "v"(d1.w), "v"(d2.x), "v"(d2.z), "v"(d3.x), "v"(d3.z));
    }
#endif
}

/// One interior-node visit: fetch its record, test both children's boxes, descend / stack / pop.
/// Precondition: state == DESCEND.
/// NEAREST (fast kernels, regular boxes): the child whose box is entered first is visited first, pruning is widened by the band
/// and odd boxes mark the query; `ordered` = false keeps this
/// lane's query in the reference's order to the letter (a Sun-shadow query, a query trav_settle sent round again). Both box
/// tests of the record count as performed.
template <bool COUNT, int BOXES = GD_BOXES_RUNTIME, bool NEAREST = false>
GD_FN void trav_step_box(const Scene &sc, const Ray &r, F3 rdiv, Trav &t, TravStack &st, WorkCounters *wc, bool ordered = true) {
    static_assert(!NEAREST || BOXES == GD_BOXES_FAST, "nearest-first walks are for trees of regular boxes");
#if GD_ADDR32
    // (the uploader refuses trees beyond 2^26 records: a record's byte offset fits 32 bits, and base + zero-extended offset is an
    // addressing mode of the load itself — no 64-bit address arithmetic per step, one address register instead of two)
    const float4 *rec = (const float4 *)((const char *)sc.recs + (uint32_t)(t.node << 6));
#else
    const float4 *rec = sc.recs + 4 * (size_t)t.node;
#endif
    float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
    // Keep the two child refs in the 16-byte loads: without this the compiler narrows the loads to 12 bytes and
    // fetches a ref with a separate, dependent 4-byte load inside the branch that needs it (one more memory
    // round trip per step).
    asm volatile("" : "+v"(q0.w), "+v"(q1.w));
#if defined(GD_TA_PROBE) || defined(GD_VALU_PROBE)
    // Measurement hooks (never defined in the product build; tools/ab_build.sh x "-DGD_TA_PROBE=2"): n more 16-byte fetches of
    // the same record line, resp. n more dependent VALU instructions, per node visit — what tells a request-bound kernel from
    // an issue-bound one (profiles/r02/vector_memory_bound.txt).
#ifdef GD_TA_PROBE
    for (int k = 0; k < GD_TA_PROBE; k++) {
        const float4 *rp = rec + (k & 3);
        asm volatile("" : "+v"(rp));
        float4 dummy = *rp;
        asm volatile("" ::"v"(dummy.x), "v"(dummy.y), "v"(dummy.z), "v"(dummy.w));
    }
#endif
#ifdef GD_VALU_PROBE
    {
        float dummy = q0.x;
#pragma unroll
        for (int k = 0; k < GD_VALU_PROBE; k++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(dummy) : "v"(q1.x));
        asm volatile("" ::"v"(dummy));
    }
#endif
#endif
    float el, eh;
    bool hl, hh;
    // (a version carrying both boxes' arithmetic in packed 2-wide vectors made the compiler turn the validity selects
    // into scalar mask logic and ran 17 % slower — tools/ab.py)
    if (BOXES == GD_BOXES_EXACT || (BOXES == GD_BOXES_RUNTIME && sc.exact_boxes)) {  // trees with an irregular box (wild input) only
        hl = aabb_entry<true>(r, rdiv, xyz(q0), xyz(q1), el);
        hh = aabb_entry<true>(r, rdiv, xyz(q2), xyz(q3), eh);
    } else if (GD_QUICK_BOXES && BOXES == GD_BOXES_FAST) {
        // the quick answer first (box_quick.h): it stands for all but a few boxes in 100 000; the lanes where one of the two is
        // withdrawn run the six face tests (a divergent region that the wave skips when no lane needs it)
        // (the ray's slack is a pure function of the ray: the compiler computes it where the ray changes — the refill —, not per step)
        const float cs = gq_ray_slack(sc.box_slack, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, rdiv.x, rdiv.y, rdiv.z);
        // A ray the slack does not vet (NaN: a direction component of exactly 0 — every Sun-shadow ray of a Sun on the horizon, an
        // axis-aligned camera ray —, a tiny origin component) has every quick answer withdrawn: a wave that holds only such rays does not
        // run the quick tests at all (wave-uniform branch), a mixed wave runs them for the lanes that can use them.
        bool sl = false, sh = false;
        if (__ballot(cs == cs) != 0) {
            sl = box_quick(r, rdiv, xyz(q0), xyz(q1), cs, el, hl);
            sh = box_quick(r, rdiv, xyz(q2), xyz(q3), cs, eh, hh);
        }
        bool ol = false, oh = false;
#ifdef GD_QUICK_CHECK
        {
            float fl, fh; bool xl, xh;
            const bool gl = aabb_entry<false, true>(r, rdiv, xyz(q0), xyz(q1), fl, &xl);
            const bool gh = aabb_entry<false, true>(r, rdiv, xyz(q2), xyz(q3), fh, &xh);
            const bool bad_l = sl & ((gl != hl) | (gl & (fl != el)) | xl), bad_h = sh & ((gh != hh) | (gh & (fh != eh)) | xh);
            const unsigned long long act = __ballot(1), need = __ballot(!(sl & sh));
            if (__ffsll((long long)act) - 1 == (int)(threadIdx.x & 63)) {
                atomicAdd(&g_quick_stats[0], 2ull * __popcll(act));
                atomicAdd(&g_quick_stats[2], 1ull);
                if (need) atomicAdd(&g_quick_stats[3], 1ull);
            }
            atomicAdd(&g_quick_stats[1], (unsigned long long)sl + (unsigned long long)sh);
            if (bad_l | bad_h) atomicAdd(&g_quick_stats[4], (unsigned long long)bad_l + (unsigned long long)bad_h);
            if ((sl & xl) | (sh & xh)) atomicAdd(&g_quick_stats[5], 1ull);
        }
#endif
        if (!(sl & sh)) {
            // (the boxes are fetched again — an L1 hit, a few times per 100 000 boxes — so that the record's 14 registers are not
            //  kept alive across the quick tests for this region's sake: k_trace has none to spare)
            const float4 *again = rec;
            asm volatile("" : "+v"(again));
            const float4 p0 = again[0], p1 = again[1], p2 = again[2], p3 = again[3];
            hl = aabb_entry<false, true>(r, rdiv, xyz(p0), xyz(p1), el, &ol);
            hh = aabb_entry<false, true>(r, rdiv, xyz(p2), xyz(p3), eh, &oh);
        }
        // a box whose entry parameter is not its slab entry: this query's answer may hinge on the visiting order (trav_settle)
        if (NEAREST && GD_CERT_ODD) t.second = (ordered & (ol | oh)) ? -__builtin_inff() : t.second;
    } else if (NEAREST && GD_CERT_ODD) {
        bool ol, oh;
        hl = aabb_entry<false, true>(r, rdiv, xyz(q0), xyz(q1), el, &ol);
        hh = aabb_entry<false, true>(r, rdiv, xyz(q2), xyz(q3), eh, &oh);
        // a box whose entry parameter is not its slab entry: this query's answer may hinge on the visiting order (trav_settle)
        t.second = (ordered & (ol | oh)) ? -__builtin_inff() : t.second;
    } else {
        hl = aabb_entry(r, rdiv, xyz(q0), xyz(q1), el);
        hh = aabb_entry(r, rdiv, xyz(q2), xyz(q3), eh);
    }
    if (COUNT) {
        wc->nodes += NEAREST ? 2 : 1;  // the lower child's box test (the upper one is counted when the reference reaches it)
        wc->steps++;
        wc->steps_top += __float_as_uint(q2.w) & 1u;  // the record's level is below GPUART_HIP_TOP_DEPTH (diagnostic)
    }
    if (NEAREST) {
        uint32_t ref_n = __float_as_uint(q0.w), ref_f = __float_as_uint(q1.w);
        float en = hl ? el : GD_ENTRY_MISS, ef = hh ? eh : GD_ENTRY_MISS;
        if (ordered && ef < en) {  // (equal entries, both origins inside included: lower child first, as the reference)
            const uint32_t rr = ref_n; ref_n = ref_f; ref_f = rr;
            const float ee = en; en = ef; ef = ee;
        }
        // (no path maxima are carried: on a path without odd boxes — and a query that meets one is walked again — entry parameters
        //  never decrease towards the leaves, see aabb_entry; a leaf's own is the largest)
        const float band = trav_band(ordered);
        const float limit = t.closest * band;
        // (a child entered beyond the limit is not stacked: the closest hit only ever comes nearer, the pop would skip it —
        //  GD_ENTRY_MISS is beyond every limit)
        if (!(ef > limit)) {
            StackEntry e;
            e.ref = ref_f; e.pe = t.entry; e.he = ef;
            st.push(e);
        }
        if (!(en > limit)) {
            trav_enter(t, ref_n, en);
            return;
        }
        trav_pop<false>(t, st, wc, true, band);
        return;
    }
    // (the counting variants stack every upper child: the reference's box test of it is counted where its walk performs it, at
    //  pop time; otherwise an upper child that is missed, or entered beyond the closest hit — which only ever comes nearer —, need
    //  not wait on the stack for a pop that would skip it)
    if (COUNT || (hh && !(eh > t.closest))) {
        StackEntry e;
        e.ref = __float_as_uint(q1.w); e.pe = t.entry; e.he = hh ? eh : GD_ENTRY_MISS;
        st.push(e);
    }
    if (hl && !(el > t.closest)) {
        trav_enter(t, __float_as_uint(q0.w), el);
        return;
    }
    trav_pop<COUNT>(t, st, wc);
}

/// Tests the primitives of the pending leaf, then pops. Precondition: state == LEAF or LEAF_TRIS.
template <bool ANY_HIT, bool COUNT, int TYPES = GD_ALL_TYPES, bool NEAREST = false>
GD_FN void trav_step_leaf(const Scene &sc, const Ray &r, Trav &t, TravStack &st, WorkCounters *wc, bool ordered = true) {
    bool stop;
    // (the triangle-mesh kernels keep the loop for their few other leaves — the floor disc —: the pair path there costs the
    // closest-hit launches 2 % for nothing)
    constexpr bool SMALL_PATH = (TYPES & 0xF) != ((1 << P_DISC) | (1 << P_TRIANGLE));
    if (t.state & 8) stop = leaf_test_tris<ANY_HIT, COUNT, NEAREST>(sc, r, t.node, t.state == TRAV_LEAF_TRIS, t, wc);
    else if (SMALL_PATH && t.state != TRAV_LEAF) stop = leaf_test_small<ANY_HIT, COUNT, TYPES, NEAREST>(sc, r, t.node, t.state == TRAV_LEAF_PAIR, t, wc);
    else stop = leaf_test<ANY_HIT, COUNT, TYPES, NEAREST>(sc, r, t.node, t, wc);
    if (stop && ANY_HIT) {
        t.state = TRAV_DONE;
        return;
    }
    trav_pop<COUNT && !NEAREST>(t, st, wc, true, NEAREST ? trav_band(ordered) : 1.0f);
}

// ---- thin-wave modes: M = 2 or 4 lanes per ray ----------------------------------------------------------------------
// A persistent wave that is draining its last rays still pays for whole instructions: a wave64 VALU instruction takes its
// issue slots and a 16-byte load instruction its >= 16 cycles in the address unit however few lanes are active, and a node
// visit is a chain of ~200 dependent instructions. When a wave holds at most 64 / M rays, every ray is therefore carried
// by M adjacent lanes as IDENTICAL replicas (same registers, same control flow, one stack column written by the first
// replica), and only the two expensive parts of a visit are split between them:
//   * the node record: replica j of a quad loads quad j of the record (one load instruction per wave instead of four),
//     tests the three planes its quad names — the minimum faces of the lower child's box, its maximum faces, and the same
//     for the upper child — against the partner's quad, and the six candidates of a box are reduced with a quad-permute
//     min. Same operands, same operations as `aabb_entry`: med3 is symmetric in its bounds, and a minimum of non-NaN
//     values does not depend on the order it is taken in (the sign of a zero entry parameter may differ; entry parameters
//     are only ever compared). A pair (M = 2) gives each lane one whole box;
//   * a leaf of one or two triangles: each half of the group tests one triangle.
// Everything else (stack, descent, pops, the generic leaf loops) runs replicated and unchanged, so the walk visits the same
// nodes in the same order and returns the same bits. Fast-form boxes only (trees with irregular boxes stay in wide mode).

/// quad_perm controls of v_mov_b32_dpp: lane i of every quad reads lane p_i of the same quad.
#define GD_QUAD_PERM(p0, p1, p2, p3) ((p0) | ((p1) << 2) | ((p2) << 4) | ((p3) << 6))
template <int CTRL>
GD_FN uint32_t quad_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
GD_FN float quad_f(float v) { return __uint_as_float(quad_u<CTRL>(__float_as_uint(v))); }

/// What a lane fetches for its ray's next step: its part of the node record — quad `sub` of it (four lanes per ray), or the box of
/// child `sub` as a = {min, lo ref}, b = {max, hi ref} (two lanes) —, or its triangle of a leaf of one or two triangles (a, b, c).
struct ThinFetch {
    float4 a, b, c;
};

/// Issues the loads for the step `t.state` announces. The pipelined loops call it as soon as a step has decided the next one, so that
/// the round trip to memory runs beside the other rays' steps; leaves other than triangle pairs fetch inside their step.
template <int M>
GD_FN void thin_fetch(const Scene &sc, const Trav &t, uint32_t sub, ThinFetch &pf) {
    static_assert(M == 2 || M == 4, "two or four lanes per ray");
    if (t.state == TRAV_DESCEND) {
        const float4 *rec = sc.recs + 4 * (size_t)t.node;
        if (M == 4) {
            pf.a = rec[sub];  // 0: lo.min | lo ref, 1: lo.max | hi ref, 2: hi.min, 3: hi.max
            asm volatile("" : "+v"(pf.a.w));  // keep the ref in the 16-byte load (see trav_step_box)
        } else {
            pf.a = rec[2 * sub]; pf.b = rec[2 * sub + 1];  // sub 0: the lower child's box (and both refs), 1: the upper child's
            asm volatile("" : "+v"(pf.a.w), "+v"(pf.b.w));
        }
    } else if (t.state & 8) {
        const bool second = (M == 4 ? (sub >> 1) : sub) != 0;
        const float4 *pa = sc.prims + 3 * (size_t)t.node + ((second && t.state == TRAV_LEAF_TRIS) ? 3 : 0);
        pf.a = pa[0]; pf.b = pa[1]; pf.c = pa[2];
    }
}

#ifndef GD_THIN_PREFETCH
#define GD_THIN_PREFETCH 1
#endif
/// Where a thin-wave step may warm the vector L1 for a child it STACKS (a pop will fetch that child's record — or a leaf's first
/// primitive record — many dependent steps later; in a draining wave that fetch is a round trip to the L2 on the ray's critical path,
/// and the memory system is idle): one 4-byte `global_load_lds` per stacked child into a sink in LDS that nobody reads — a load without
/// a destination register, so nothing has to stay alive for it. `sink` == nullptr: no prefetch.
struct ThinPrefetch {
    const float4 *recs, *prims;
    uint32_t *sink;  ///< LDS, BLOCK words of this wave
};

/// One interior-node visit of a ray carried by M replicas (`sub` = this lane's index among them), on the parts of the record the
/// replicas hold in `pf` (thin_fetch). Precondition: DESCEND.
template <int M, bool NEAREST = false>
GD_FN void trav_step_box_thin_on(F3 ro, F3 rd, F3 rdiv, Trav &t, TravStack &st, uint32_t sub, const ThinFetch &pf, float slack, bool ordered = true,
                                 ThinPrefetch warm = ThinPrefetch{nullptr, nullptr, nullptr}) {
    const float INF = __builtin_inff();
    float e;  // this lane's box: entry parameter, GD_ENTRY_MISS when not hit
    uint32_t ref_lo, ref_hi;
    if (M == 4) {
        const float4 mine = pf.a;
        const F3 other = f3(quad_f<GD_QUAD_PERM(1, 0, 3, 2)>(mine.x), quad_f<GD_QUAD_PERM(1, 0, 3, 2)>(mine.y), quad_f<GD_QUAD_PERM(1, 0, 3, 2)>(mine.z));
        const bool inside = within(ro.x, mine.x, other.x) & within(ro.y, mine.y, other.y) & within(ro.z, mine.z, other.z);
        const float kx = (mine.x - ro.x) * rdiv.x, ky = (mine.y - ro.y) * rdiv.y, kz = (mine.z - ro.z) * rdiv.z;
        const float cx = face_candidate(kx, ro.y, rd.y, mine.y, other.y, ro.z, rd.z, mine.z, other.z);
        const float cy = face_candidate(ky, ro.x, rd.x, mine.x, other.x, ro.z, rd.z, mine.z, other.z);
        const float cz = face_candidate(kz, ro.x, rd.x, mine.x, other.x, ro.y, rd.y, mine.y, other.y);
        const float half = fminf(fminf(cx, cy), cz);
        const float best = fminf(half, quad_f<GD_QUAD_PERM(1, 0, 3, 2)>(half));
        e = inside ? -1.0f : fminf(best, 1.0e+19f);
        e = (inside | (best < INF)) ? e : GD_ENTRY_MISS;
        if (NEAREST) {  // aabb_entry's `odd`: the box is entered beyond its slab entry (the partner lane holds the other plane of every axis)
            const float slab = fmaxf(fmaxf(fminf(kx, quad_f<GD_QUAD_PERM(1, 0, 3, 2)>(kx)), fminf(ky, quad_f<GD_QUAD_PERM(1, 0, 3, 2)>(ky))),
                                     fminf(kz, quad_f<GD_QUAD_PERM(1, 0, 3, 2)>(kz)));
            const uint32_t odd = (!inside & (best < INF) & (best != slab)) ? 1u : 0u;
            if (ordered & ((odd | quad_u<GD_QUAD_PERM(2, 3, 0, 1)>(odd)) != 0)) t.second = -INF;
        }
        ref_lo = quad_u<GD_QUAD_PERM(0, 0, 0, 0)>(__float_as_uint(mine.w));
        ref_hi = quad_u<GD_QUAD_PERM(1, 1, 1, 1)>(__float_as_uint(mine.w));
    } else {
        const float4 bmin = pf.a, bmax = pf.b;
        float pos;
        bool hit;
        bool odd_box = false;
        if (GD_QUICK_BOXES) {  // a pair: each lane one whole box — the quick answer first, as in trav_step_box (`slack`: Scene::box_slack)
            const float cs = gq_ray_slack(slack, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, rdiv.x, rdiv.y, rdiv.z);
            if (!box_quick(Ray{ro, rd}, rdiv, xyz(bmin), xyz(bmax), cs, pos, hit)) hit = aabb_entry<false, true>(Ray{ro, rd}, rdiv, xyz(bmin), xyz(bmax), pos, &odd_box);
        } else if (NEAREST)
            hit = aabb_entry<false, true>(Ray{ro, rd}, rdiv, xyz(bmin), xyz(bmax), pos, &odd_box);
        else
            hit = aabb_entry(Ray{ro, rd}, rdiv, xyz(bmin), xyz(bmax), pos);
        if (NEAREST) {
            const uint32_t odd = odd_box ? 1u : 0u;
            if (ordered & ((odd | quad_u<GD_QUAD_PERM(1, 0, 3, 2)>(odd)) != 0)) t.second = -INF;
        }
        e = hit ? pos : GD_ENTRY_MISS;
        ref_lo = quad_u<GD_QUAD_PERM(0, 0, 2, 2)>(__float_as_uint(bmin.w));
        ref_hi = quad_u<GD_QUAD_PERM(0, 0, 2, 2)>(__float_as_uint(bmax.w));
    }
    float el = M == 4 ? quad_f<GD_QUAD_PERM(0, 0, 0, 0)>(e) : quad_f<GD_QUAD_PERM(0, 0, 2, 2)>(e);
    float eh = M == 4 ? quad_f<GD_QUAD_PERM(2, 2, 2, 2)>(e) : quad_f<GD_QUAD_PERM(1, 1, 3, 3)>(e);
    if (NEAREST && ordered && eh < el) {  // the upper child's box is entered first (GD_ENTRY_MISS is the largest value): it goes first
        const uint32_t rr = ref_lo; ref_lo = ref_hi; ref_hi = rr;
        const float ee = el; el = eh; eh = ee;
    }
    const float band = NEAREST ? trav_band(ordered) : 1.0f;
    const bool writer = sub == 0;
    const float limit = t.closest * band;
    if (!(eh > limit)) {  // (as trav_step_box: what a pop would skip is not stacked; GD_ENTRY_MISS is beyond every limit)
        StackEntry s;
        s.ref = ref_hi; s.pe = t.entry; s.he = eh;
        st.push(s, writer);
        if (GD_THIN_PREFETCH && warm.sink && writer && !(el > limit)) {  // (if the other child is not entered either, this one is popped at once)
            const float4 *line = (ref_hi & GD_REF_LEAF) ? warm.prims + 3 * (size_t)(ref_hi & GD_REF_INDEX) : warm.recs + 4 * (size_t)ref_hi;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)line, (__attribute__((address_space(3))) void *)warm.sink, 4, 0, 0);
        }
    }
    if (!(el > limit)) {
        trav_enter(t, ref_lo, el);
        return;
    }
    trav_pop<false>(t, st, nullptr, writer, band);
}

/// The same, fetching the record itself (the loops that do not fetch ahead).
template <int M, bool NEAREST = false>
GD_FN void trav_step_box_thin(const Scene &sc, F3 ro, F3 rd, F3 rdiv, Trav &t, TravStack &st, uint32_t sub, bool ordered = true, uint32_t *sink = nullptr) {
    ThinFetch pf;
    thin_fetch<M>(sc, t, sub, pf);
    trav_step_box_thin_on<M, NEAREST>(ro, rd, rdiv, t, st, sub, pf, sc.box_slack, ordered, ThinPrefetch{sc.recs, sc.prims, sink});
}

/// The pending leaf of a ray carried by M replicas, then the pop. A leaf of one or two triangles is split between the halves
/// of the group (its triangle in `pf`, thin_fetch); every other leaf runs replicated through the code of `trav_step_leaf`.
template <int M, int TYPES, bool NEAREST = false>
GD_FN void trav_step_leaf_thin_on(const Scene &sc, F3 ro, F3 rd, Trav &t, TravStack &st, uint32_t sub, const ThinFetch &pf, bool ordered = true) {
    const bool writer = sub == 0;
    constexpr bool SMALL_PATH = (TYPES & 0xF) != ((1 << P_DISC) | (1 << P_TRIANGLE));
    const Ray r{ro, rd};
    if (t.state & 8) {
        const bool two = t.state == TRAV_LEAF_TRIS;
        const float tt = triangle_t(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, pf.a, pf.b, pf.c);
        const float ta = M == 4 ? quad_f<GD_QUAD_PERM(0, 0, 0, 0)>(tt) : quad_f<GD_QUAD_PERM(0, 0, 2, 2)>(tt);
        float tb = M == 4 ? quad_f<GD_QUAD_PERM(2, 2, 2, 2)>(tt) : quad_f<GD_QUAD_PERM(1, 1, 3, 3)>(tt);
        tb = two ? tb : -1.0f;
        take_hit<NEAREST>(t, ta, t.node);
        take_hit<NEAREST>(t, tb, t.node + 1);
    } else if (SMALL_PATH && t.state != TRAV_LEAF) {
        leaf_test_small<false, false, TYPES, NEAREST>(sc, r, t.node, t.state == TRAV_LEAF_PAIR, t, nullptr);
    } else {
        leaf_test<false, false, TYPES, NEAREST>(sc, r, t.node, t, nullptr);
    }
    trav_pop<false>(t, st, nullptr, writer, NEAREST ? trav_band(ordered) : 1.0f);
}

template <int M, int TYPES, bool NEAREST = false>
GD_FN void trav_step_leaf_thin(const Scene &sc, F3 ro, F3 rd, Trav &t, TravStack &st, uint32_t sub, bool ordered = true) {
    ThinFetch pf;
    thin_fetch<M>(sc, t, sub, pf);  // (a state that is not a triangle leaf fetches nothing here)
    trav_step_leaf_thin_on<M, TYPES, NEAREST>(sc, ro, rd, t, st, sub, pf, ordered);
}

/// Runs one query to completion (megakernels and test hooks).
/// NEAREST: the fast kernels' walk (the caller guarantees regular boxes) including its second walk in the reference's order
/// where trav_settle asks for one — for the test hook that pins that walk to the fixtures.
template <bool ANY_HIT, bool COUNT, bool NEAREST = false>
GD_FN void traverse(const Scene &sc, const Ray &r, TravStack &st, float &closest, uint32_t &hit_prim, WorkCounters *wc) {
    F3 rdiv = f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
    Trav t;
    constexpr int BOXES = NEAREST ? GD_BOXES_FAST : GD_BOXES_RUNTIME;
    bool ordered = true;
    for (;;) {
        trav_init<BOXES>(sc, r, rdiv, t, st, wc, COUNT && ordered, NEAREST && ordered);
        while (t.state != TRAV_DONE) {
            if (t.state == TRAV_DESCEND) trav_step_box<COUNT, BOXES, NEAREST>(sc, r, rdiv, t, st, wc, ordered);
            else trav_step_leaf<ANY_HIT, COUNT, GD_ALL_TYPES, NEAREST>(sc, r, t, st, wc, ordered);
        }
        if (!trav_settle<NEAREST>(t, ordered)) break;
        ordered = false;
    }
    closest = t.closest;
    hit_prim = t.hit_prim;
}

/// Recomputes point, normal and type of the winning primitive (pure function of ray + record).
GD_FN void shade_prim(const Scene &sc, const Ray &r, uint32_t pi, Surface &s) {
    float4 q0 = sc.prims[3 * pi], q1 = sc.prims[3 * pi + 1], q2 = sc.prims[3 * pi + 2];
#ifdef GD_SHADE_PROBE
    // Measurement hook (never in the product build): the hit primitive's three quads are fetched GD_SHADE_PROBE more times — what
    // carrying them over from the query instead of re-fetching them could at best save: nothing measurable
    // (profiles/r03/request_trims.txt)
    for (int k = 0; k < GD_SHADE_PROBE; k++) {
        const float4 *rp = sc.prims + 3 * (size_t)pi;
        asm volatile("" : "+v"(rp));
        float4 d0 = rp[0], d1 = rp[1], d2 = rp[2];
        asm volatile("" ::"v"(d0.x), "v"(d0.w), "v"(d1.x), "v"(d1.w), "v"(d2.x), "v"(d2.w));
    }
#endif
    prim_hit(r, q0, q1, q2, s.pos, s.p, s.n, s.ptype);
}

/// reference CheckIntersectionInclUserSphere (shaders/intersection.glsl:71-111) on top of a
/// finished BVH query (closest, hit_prim).
GD_FN void resolve_hit(const Scene &sc, const Ray &r, float closest, uint32_t hit_prim, const float us[4], Surface &s,
                       bool &user_sphere_hit) {
    if (hit_prim != GD_NO_PRIM) shade_prim(sc, r, hit_prim, s);
    else { s.pos = -1; s.ptype = -1; }
    (void)closest;
    float usPos; F3 usP = f3(0, 0, 0), usN = f3(0, 0, 0);
    sphere_hit(r, f3(us[0], us[1], us[2]), us[3], usPos, usP, usN);
    if (usPos > GD_VISIBILITY_OFFSET && (s.pos < 0 || usPos < s.pos)) {
        user_sphere_hit = true;
        s.ptype = P_SPHERE; s.pos = usPos; s.p = usP; s.n = usN;
    } else
        user_sphere_hit = false;
}

}  // namespace gd
