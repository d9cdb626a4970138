// device_scene.h — device-side scene layout, primitive intersectors and BVH traversal.
//
// Device layout (built by the uploader from the reference's canonical quad array):
//   nodes : float4[2*N], node n (pre-order ordinal; the lower child of n is n+1):
//             [2n]   = { bbmin.xyz, bits(link) }   link = hi-child ordinal (interior) |
//                                                         first primitive index (leaf)
//             [2n+1] = { bbmax.xyz, bits(meta) }   meta = bit31 leaf | primitive count
//   prims : float4[3*P], fixed 48-byte records, primitive p at [3p..3p+2]:
//             sphere   { c.xyz, T }{ r, 0, 0, 0 }{ 0 }
//             disc     { c.xyz, T }{ n.xyz, r }{ 0 }
//             triangle { v0.xyz, T }{ e1.xyz, 0 }{ e2.xyz, 0 }     e1 = v1-v0, e2 = v2-v0 (fp32)
//             cone     { c1.xyz, T }{ axis.xyz, len }{ r1, widthCoeff, cosB, dotAxC1 }
//           T = bits(type)
//   parent: uint32[N] = parent ordinal | bit31 "is lower child" (only used by the stackless
//           fallback for trees deeper than the LDS stack).
#pragma once
#include "device_math.h"

namespace gd {

enum { P_SPHERE = 0, P_DISC = 1, P_TRIANGLE = 2, P_CONE = 3 };
#define GD_VISIBILITY_OFFSET 1.0e-4f
#define GD_NO_PRIM 0xffffffffu
#define GD_USER_SPHERE 0xfffffffeu
#define GD_META_LEAF 0x80000000u

struct Scene {
    const float4 *__restrict__ nodes;
    const float4 *__restrict__ prims;
    const uint32_t *__restrict__ parent;
    uint32_t num_nodes;
    uint32_t max_depth;
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

/// One primitive record against a ray (reference CheckBVHPrimitiveIntersection,
/// shaders/bvh_intersection.glsl:125-223, including its `pos < VISIBILITY_OFFSET -> -1` cut).
GD_FN void prim_hit(const Ray &r, float4 q0, float4 q1, float4 q2, float &pos, F3 &p, F3 &n, int &ptype) {
    ptype = (int)__float_as_uint(q0.w);
    if (ptype == P_TRIANGLE) triangle_hit(r, xyz(q0), xyz(q1), xyz(q2), pos, p, n);
    else if (ptype == P_SPHERE) sphere_hit(r, xyz(q0), q1.x, pos, p, n);
    else if (ptype == P_DISC) disc_hit(r, xyz(q0), q1.w, xyz(q1), pos, p, n);
    else cone_hit(r, xyz(q0), q2.x, xyz(q1), q1.w, q2.y, q2.z, q2.w, pos, p, n);
    if (pos < GD_VISIBILITY_OFFSET) pos = -1;
}

/// Ray/AABB entry test (reference IntersectsAABB, shaders/bvh_intersection.glsl:229-354).
/// Returns false on a miss; pos = -1 when the origin is inside (inclusive), else the smallest
/// non-negative plane parameter whose hit point lies within the face (inclusive bounds).
GD_FN bool aabb_entry(const Ray &r, F3 rdiv, F3 bmin, F3 bmax, float &pos) {
    if (r.o.x >= bmin.x && r.o.y >= bmin.y && r.o.z >= bmin.z && r.o.x <= bmax.x && r.o.y <= bmax.y && r.o.z <= bmax.z) {
        pos = -1;
        return true;
    }
    bool hit = false;
    pos = 1.0e+19f;
#define GD_FACE(K, A0, B0, A1, B1, LOA, HIA, LOB, HIB)                         \
    {                                                                          \
        float k = (K);                                                         \
        if (k >= 0) {                                                          \
            float a = (A0) + k * (A1), b = (B0) + k * (B1);                    \
            if (a >= (LOA) && a <= (HIA) && b >= (LOB) && b <= (HIB)) {        \
                hit = true;                                                    \
                if (k < pos) pos = k;                                          \
            }                                                                  \
        }                                                                      \
    }
    if (r.d.x != 0) {
        GD_FACE((bmin.x - r.o.x) * rdiv.x, r.o.y, r.o.z, r.d.y, r.d.z, bmin.y, bmax.y, bmin.z, bmax.z)
        GD_FACE((bmax.x - r.o.x) * rdiv.x, r.o.y, r.o.z, r.d.y, r.d.z, bmin.y, bmax.y, bmin.z, bmax.z)
    }
    if (r.d.y != 0) {
        GD_FACE((bmin.y - r.o.y) * rdiv.y, r.o.x, r.o.z, r.d.x, r.d.z, bmin.x, bmax.x, bmin.z, bmax.z)
        GD_FACE((bmax.y - r.o.y) * rdiv.y, r.o.x, r.o.z, r.d.x, r.d.z, bmin.x, bmax.x, bmin.z, bmax.z)
    }
    if (r.d.z != 0) {
        GD_FACE((bmin.z - r.o.z) * rdiv.z, r.o.x, r.o.y, r.d.x, r.d.y, bmin.x, bmax.x, bmin.y, bmax.y)
        GD_FACE((bmax.z - r.o.z) * rdiv.z, r.o.x, r.o.y, r.d.x, r.d.y, bmin.x, bmax.x, bmin.y, bmax.y)
    }
#undef GD_FACE
    return hit;
}

/// Tests the `count` primitives of a leaf; keeps the strictly closer hit (first one wins ties,
/// reference shaders/bvh_intersection.glsl:405-423). Returns true if ANY_HIT and something was hit.
template <bool ANY_HIT, bool COUNT>
GD_FN bool leaf_test(const Scene &sc, const Ray &r, uint32_t first, uint32_t count, float &closest, uint32_t &hit_prim,
                     WorkCounters *wc) {
    for (uint32_t i = 0; i < count; i++) {
        uint32_t pi = first + i;
        float4 q0 = sc.prims[3 * pi], q1 = sc.prims[3 * pi + 1], q2 = sc.prims[3 * pi + 2];
        float pos; F3 p, n; int ptype;
        prim_hit(r, q0, q1, q2, pos, p, n, ptype);
        if (COUNT) wc->prims[ptype & 3]++;
        if (pos > 0 && pos < closest) {
            closest = pos;
            hit_prim = pi;
            if (ANY_HIT) return true;
        }
    }
    return false;
}

/// Closest-hit query, LDS-stack form. Visits exactly the nodes, in exactly the order, of the
/// reference's stackless parent-pointer walk (shaders/bvh_intersection.glsl:360-457): lower child
/// first, prune on `entry > closest`, and — where the reference re-tests a parent's box when it
/// returns from the lower child — the parent's entry parameter kept on the stack is compared with
/// the current closest hit instead (same value, since the test is a pure function of node and ray).
/// Stack entries: (hi-child ordinal, parent entry parameter), one column per lane.
template <bool ANY_HIT, bool COUNT, int STACK_DEPTH, int BLOCK>
GD_FN void traverse_stack(const Scene &sc, const Ray &r, uint2 (*stack)[BLOCK], int lane, float &closest,
                          uint32_t &hit_prim, WorkCounters *wc) {
    F3 rdiv = f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
    closest = 1e+19f;
    hit_prim = GD_NO_PRIM;
    uint32_t node = 0;
    int sp = 0;
    if (COUNT) wc->rays++;
    for (;;) {
        float4 n0 = sc.nodes[2 * node], n1 = sc.nodes[2 * node + 1];
        if (COUNT) wc->nodes++;
        float entry;
        bool hit = aabb_entry(r, rdiv, xyz(n0), xyz(n1), entry);
        if (hit && !(entry > closest)) {
            uint32_t meta = __float_as_uint(n1.w), link = __float_as_uint(n0.w);
            if (meta & GD_META_LEAF) {
                if (leaf_test<ANY_HIT, COUNT>(sc, r, link, meta & ~GD_META_LEAF, closest, hit_prim, wc) && ANY_HIT) return;
            } else {
                stack[sp++][lane] = make_uint2(link, __float_as_uint(entry));
                node = node + 1;
                continue;
            }
        }
        // return towards the root until a pending upper child is still worth visiting
        for (;;) {
            if (sp == 0) return;
            uint2 e = stack[--sp][lane];
            if (__uint_as_float(e.y) > closest) continue;
            node = e.x;
            break;
        }
    }
}

/// The same query as a literal parent-pointer walk (no stack), for trees deeper than the LDS stack.
template <bool ANY_HIT, bool COUNT>
GD_FN void traverse_stackless(const Scene &sc, const Ray &r, float &closest, uint32_t &hit_prim, WorkCounters *wc) {
    F3 rdiv = f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
    closest = 1e+19f;
    hit_prim = GD_NO_PRIM;
    uint32_t node = 0;
    bool returning = false;
    int from = 0;  // 0 none, 1 lower, 2 upper
    if (COUNT) wc->rays++;
    for (;;) {
        if (returning && from == 2 && node == 0) return;
        float4 n0 = sc.nodes[2 * node], n1 = sc.nodes[2 * node + 1];
        uint32_t meta = __float_as_uint(n1.w), link = __float_as_uint(n0.w);
        if (COUNT && !returning) wc->nodes++;
        float entry;
        if (aabb_entry(r, rdiv, xyz(n0), xyz(n1), entry)) {
            if (entry > closest)
                returning = true;
            else if (meta & GD_META_LEAF) {
                if (leaf_test<ANY_HIT, COUNT>(sc, r, link, meta & ~GD_META_LEAF, closest, hit_prim, wc) && ANY_HIT) return;
                returning = true;
            } else {
                returning = false;
                if (from == 0) node = node + 1;
                else if (from == 1) { from = 0; node = link; }
                else returning = true;
            }
        } else
            returning = true;
        if (returning) {
            uint32_t pw = sc.parent[node];
            from = (pw & 0x80000000u) ? 1 : 2;
            if (node == 0) from = 2;
            node = pw & 0x7fffffffu;
        }
    }
}

/// Recomputes point, normal and type of the winning primitive (pure function of ray + record).
GD_FN void shade_prim(const Scene &sc, const Ray &r, uint32_t pi, Surface &s) {
    float4 q0 = sc.prims[3 * pi], q1 = sc.prims[3 * pi + 1], q2 = sc.prims[3 * pi + 2];
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
