// box_quick.h — a cheap ray / box answer that is EITHER certainly the reference's (IntersectsAABB,
// shaders/bvh_intersection.glsl:229-354, bit for bit) OR withdrawn; the caller then runs the six face tests (aabb_entry).
//
// Why. The reference evaluates the ray at each of the six plane parameters and checks the point against the face: ~95 VALU
// instructions per box, 190 of the ~266 of a traversal step (profiles/r04/slab_box_test_probe.txt). In exact arithmetic the
// answer is the slab entry — the largest of the three nearer-plane parameters, if it does not exceed the smallest of the farther
// ones. In fp32 the two differ exactly where a parameter of one axis comes within rounding of a plane of another (rays through
// edges and corners, origins on a plane). This file decides from the six plane parameters alone, with margins that cover every
// rounding of the reference's own expressions, whether that can be the case — 46 VALU — and withdraws when it can.
//
// Notation, per axis a: o, d the ray's origin and direction component, r = fl(1/d) (the caller's rdiv), lo <= hi the box planes,
//   k_lo = fl(fl(lo - o) * r), k_hi likewise       the reference's plane parameters (:258-351), bit for bit what aabb_entry computes
//   near = min(k_lo, k_hi), far = max(k_lo, k_hi)   (the plane the ray meets first / last; fl(p - o) * r is monotone in p)
//   A(k) = fl(o + fl(k * d))                        the coordinate the reference tests when a face of ANOTHER axis has parameter k
//   u = 2^-24; arithmetic is round-to-nearest with denormals flushed (inputs and results), as the product and llvmpipe run.
// Margins: M(k) = RHO * |k| + E_a,  E_a = cs * |r_a|,  RHO = 10 u,  cs = 4 u * Pmax + 2^-90, Pmax >= every |plane| of the tree (<= 2^40);
//          m(k) = RHO * |k| + TAU, TAU = 2^-50, where only Lemma P is needed (it has no term in Pmax).
// Rays: gq_ray_slack hands out the tree's cs only to a ray with |o_a| < 2^40 that is 0 or at least 2^-60, 2^-60 < |d_a| < 2^20 (so
// 2^-20 < |r_a| < 2^60) and no NaN — every camera, bounce and shadow ray of a sane scene; any other ray gets NaN and every answer for
// it is withdrawn (all comparisons below are false, and the inside answer is gated by cs == cs). For the rays that remain every plane
// parameter is finite, |k| < 2^101, and never NaN. Trees: every plane coordinate is 0 or at least 2^-60 in magnitude (else cs = +inf).
//
// Lemma F (a face certainly FAILS its a-check).  k any float.
//   k <= near - M(near)  =>  A(k) lies strictly before the nearer plane:  for d > 0, A(k) < lo.
//   k >= far  + M(far)   =>  A(k) lies strictly beyond the farther plane: for d > 0, A(k) > hi.     (d < 0: lo and hi swapped)
//   Proof (d > 0, first line). s = fl(lo - o) = (lo - o)(1 + e1) + f1, near = s r (1 + e2) + f2, r = (1 + e0) / d, x = fl(k d) =
//   k d (1 + e3) + f3 with |e| <= u and flush errors |f| <= 2^-126. Hence near * d = (lo - o)(1 + t), |t| <= 3.1 u, up to (|f1| + |f2| d),
//   and with k <= near - D:  o + x <= lo + 4.2 u |lo - o| - D d (1 - u) + 2^-126 (2 + d) (1 + u).  A(k) = fl(o + x) < lo holds as soon as
//   o + x <= pred(lo), and pred(lo) >= lo - 2 u |lo| - 2^-125 (flushing included). So D >= 4.3 u |near| + (2.1 u |lo| + 2^-124) |r| + 2^-125
//   suffices. What the code subtracts covers it with the roundings of its own two instructions (x = fl(near - E), U = fl(x - RHO |x|),
//   relative error u each): where |near| <= 8 E, U <= near - E (1 - 18 u) and the requirement is at most near - (0.55 + 35 u) E - 2^-125
//   (2.1 u Pmax |r| = 0.525 E; 2^-124 |r| and 2^-125 vanish beside E >= 2^-90 * 2^-20); elsewhere |x| >= 7/8 |near| (1 - u), so
//   U <= near - E - (0.875 RHO - 2.25 u) |near| = near - E - 6.5 u |near|.
//   An overflowing k * d is +-inf on the right side. The second line and d < 0 are the same computation mirrored.
// Lemma P (a face certainly PASSES its a-check).  near + 4.3 u |near| + 2^-63 <= k <= far - 4.3 u |far| - 2^-63  =>  lo <= A(k) <= hi.
//   Proof: as above without the term 2 u |lo|: real o + x >= lo suffices, since rounding to nearest is monotone and lo is a float (a
//   flushed result is 0, which lies on the right side of a bound of the other sign); the flush terms are (2^-124 |r| + 2^-125) < 2^-63.
//   What the code compares is fl(k0 + RHO |k0|) <= fl(T - TAU) for a k0 >= near (and fl(T + TAU) <= fl(X - RHO |X|) for an X <= far):
//   x + 4.3 u |x| grows with x; fl(k0 + RHO |k0|) exceeds k0 + 4.3 u |k0| by 3.7 u |k0| at least; fl(T - TAU) <= T always and
//   <= T - TAU / 2 while T <= 2^-27; and beyond that either |k0| >= 2^-40, where 3.7 u |k0| > 2^-63, or k0 + 4.3 u |k0| + 2^-63 < 2^-39 < T.
//
// With U = max_a (near_a - M_a), V = min_a (far_a + M_a):
//  MISS  <=  not inside, max(U, 0) >= V.   Every face has a parameter k; k < 0 or NaN fails. Let e attain V, a attain U. If 0 >= V_e:
//        far_e < 0, both faces of axis e fail on k, every other face has k >= 0 >= V_e and fails its e-check (F). Else U_a >= V_e
//        (a != e because U_a < near_a <= far_a < V_a): k >= V_e fails the e-check, k < V_e <= U_a fails the a-check; the faces of
//        axis e have k <= far_e < V_e <= U_a (a-check fails), those of axis a have k >= near_a > U_a >= V_e (e-check fails).
//  CLEAN HIT at T = max_a near_a = near_c  <=  not inside,
//        T >= 0,  grow(m2) <= T - TAU,  T + TAU <= shrink(min_a far_a),  m2 <= U,   m2 = med3_a(near_a) = the largest near_a, a != c.
//        The entry face (axis c, nearer plane) passes: its parameter is T >= 0, and for a != c Lemma P applies with k0 = m2 >= near_a
//        and X = min far <= far_a (a second axis with near_a = T makes m2 = T and the condition false). The entry axis itself is
//        among the far_a: a box that is flat on it withdraws. No other face that passes has a smaller parameter: a nearer face of a != c has
//        near_a <= med3(near) <= U, and U can only reach med3(near) when it is attained at c (near_b - M_b < near_b <= med3 for
//        b != c), so near_a <= near_c - M_c and its c-check fails (F); every farther face has far_a >= T. So the reference's running
//        minimum ends at T: hit, entry parameter min(T, 1e19) like aabb_entry, and the box is not `odd` (T is its slab entry).
//  INSIDE (the reference's inclusive test, :245-250: lo <= o <= hi on every axis)  <=>  T <= 0 <= min_a far_a, for vetted rays and trees:
//        p - o is 0 or at least 2^-84 in magnitude (both are 0 or at least 2^-60), so s = fl(p - o) has the sign of p - o and is not
//        flushed; |s r| >= 2^-104 is not flushed either, so k = fl(s r) has the sign of s times the sign of r (and k = +-0 for s = 0).
//        With r > 0: k_lo <= 0 <= k_hi <=> lo <= o <= hi, and k_lo = near, k_hi = far; with r < 0 the two swap. (+0 and -0 compare equal.)
// (shrink / grow are monotone, so they are applied once, after the reduction over the axes.)
//
// Checked: tools/quick_box_check.cpp (this very file on the CPU under FTZ / DAZ against the reference's comparison form: adversarial
// rays through edges, corners, planes; flat, nested, huge and tiny boxes) and -DGD_QUICK_CHECK device builds (every product box
// test both ways, mismatches counted) — profiles/r04/quick_box_test.txt.
#pragma once

#ifndef GQ_FN
#error "the includer defines GQ_FN (function attributes) and gq_min / gq_max / gq_med3 / gq_fma / gq_abs"
#endif

#ifndef GQ_RHO
#define GQ_RHO 5.9604644775390625e-07f  // 10 u = 10 * 2^-24
#endif
#ifndef GQ_TAU
#define GQ_TAU 8.8817841970012523e-16f  // 2^-50
#endif

/// The slack constant of a tree whose box planes are all finite, 0 or at least 2^-60 in magnitude (gq_plane_ok) and at most
/// `pmax` <= 2^40; +inf (every answer withdrawn) otherwise. Host side (converter / upload) and checker.
#ifndef GQ_HOST_FN
#define GQ_HOST_FN static inline
#endif
GQ_HOST_FN bool gq_plane_ok(float p) { return p == 0.0f || (p >= 8.6736173798840355e-19f || p <= -8.6736173798840355e-19f); }  // (false for NaN)
GQ_HOST_FN float gq_slack_of_tree(float pmax) {
    if (!(pmax >= 0.0f) || !(pmax <= 1.099511627776e12f)) return __builtin_inff();
    return 2.384185791015625e-07f * pmax + 8.0779356694631609e-28f;  // 4 u * Pmax + 2^-90
}

/// Per ray (a pure function of the ray: the compiler computes it where the ray changes, not per step): the tree's slack for a ray
/// the lemmas cover, NaN for any other — a component of the origin at or beyond 2^40 or within 2^-60 of zero without being zero, of
/// the direction at or beyond 2^20 or within 2^-60 of zero (rdiv at or beyond 2^60, +-inf for 0), a NaN anywhere. rx, ry, rz: the
/// caller's rdiv = 1 / d.
GQ_FN float gq_ray_slack(float cs, float ox, float oy, float oz, float dx, float dy, float dz, float rx, float ry, float rz) {
    const float wd = gq_max(gq_max(gq_abs(dx), gq_abs(dy)), gq_abs(dz)) * 3.2451855365842673e+32f;  // * 2^108: inf from 2^20 on
    const float wr = gq_max(gq_max(gq_abs(rx), gq_abs(ry)), gq_abs(rz)) * 2.9514790517935283e+20f;  // * 2^68:  inf from 2^60 on
    const float wo = gq_max(gq_max(gq_abs(ox), gq_abs(oy)), gq_abs(oz)) * 3.0948500982134507e+26f;  // * 2^88:  inf from 2^40 on
    const float nn = ((dx + dy) + dz) + ((ox + oy) + oz);                                           // NaN if any of them is (max drops NaNs)
    const float TINY = 8.6736173798840355e-19f;                                                     // 2^-60: an origin component below it that is not 0
    const float tx = gq_abs(ox) < TINY ? gq_abs(ox) : 0.0f, ty = gq_abs(oy) < TINY ? gq_abs(oy) : 0.0f, tz = gq_abs(oz) < TINY ? gq_abs(oz) : 0.0f;
    const float c = gq_fma(nn, 0.0f, gq_fma((wd + wr) + wo, 0.0f, cs));                            // 0 * inf = NaN
#ifdef GQ_NO_TINY_ORIGIN_GUARD  // teeth test of tools/quick_box_check.cpp only: without this guard the inside answer must go wrong
    (void)tx; (void)ty; (void)tz;
    return c;
#else
    return (tx + ty) + tz > 0.0f ? __builtin_nanf("") : c;
#endif
}

/// k0..k5: the six plane parameters in aabb_entry's order (x lo, x hi, y lo, y hi, z lo, z hi). ax, ay, az = |rdiv|. cs: gq_ray_slack.
/// Returns whether the answer stands; then `hit` and `pos` (-1 inside, else the entry parameter; 1e19 on a miss, like aabb_entry) are
/// the reference's. Otherwise both are unspecified.
GQ_FN bool gq_box(float k0, float k1, float k2, float k3, float k4, float k5, float ax, float ay, float az, float cs, float &pos, bool &hit) {
    const float nx = gq_min(k0, k1), fx = gq_max(k0, k1);
    const float ny = gq_min(k2, k3), fy = gq_max(k2, k3);
    const float nz = gq_min(k4, k5), fz = gq_max(k4, k5);
    const float T = gq_max(gq_max(nx, ny), nz);
    const float m2 = gq_med3(nx, ny, nz);
    const float X = gq_min(gq_min(fx, fy), fz);
    const float ncs = -cs;
    float U = gq_max(gq_max(gq_fma(ax, ncs, nx), gq_fma(ay, ncs, ny)), gq_fma(az, ncs, nz));
    U = gq_fma(-GQ_RHO, gq_abs(U), U);
    float V = gq_min(gq_min(gq_fma(ax, cs, fx), gq_fma(ay, cs, fy)), gq_fma(az, cs, fz));
    V = gq_fma(GQ_RHO, gq_abs(V), V);
    const float NI = gq_fma(GQ_RHO, gq_abs(m2), m2);
    const float FI = gq_fma(-GQ_RHO, gq_abs(X), X);
    const bool inside = (T <= 0.0f) & (X >= 0.0f) & (cs == cs);
    const bool miss = gq_max(U, 0.0f) >= V;
    const bool clean = (T >= 0.0f) & (NI <= T - GQ_TAU) & (T + GQ_TAU <= FI) & (m2 <= U);
    hit = inside | clean;
    pos = inside ? -1.0f : hit ? gq_min(T, 1.0e+19f) : 1.0e+19f;
    return hit | miss;
}
