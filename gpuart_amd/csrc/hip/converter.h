// converter.h — host side of gpuart_hip_upload_bvh: validates the reference's canonical compiled tree (RGBA32F quads,
// reference src/bvh.cpp:161-222) and re-lays it out as 64-byte two-children records + 48-byte primitive records
// (DESIGN.md section 3). Host code only; included by gpuart_hip.hip.
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "device_scene.h"

namespace {

using namespace gd;

// ---- canonical tree -> device layout ---------------------------------------------------------------
/// std::allocator whose value-less construct() leaves trivial elements uninitialised: the device arrays are touched for the first
/// time by the threads that fill them (every element is written), not zeroed by one thread beforehand.
template <class T>
struct UninitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = UninitAlloc<U>; };
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) { ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};

struct Converter {
    const float *q;
    size_t nq;
    std::vector<float4, UninitAlloc<float4>> recs, prims;
    size_t num_nodes = 0;
    uint32_t type_mask = 0;
    uint32_t max_depth = 0;
    uint32_t top_depth = 0;  ///< records of nodes above this level are marked (pad word of the record; counters only)
    std::string err;

    static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
    static float fbits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

    /// Appends the device record of one canonical primitive payload (type + data quads).
    static bool pack_prim(uint32_t type, const float *d, float4 rec[3]) {
        float T = fbits(type);
        switch (type) {
        case P_SPHERE:
            rec[0] = make_float4(d[0], d[1], d[2], T); rec[1] = make_float4(d[3], 0, 0, 0); rec[2] = make_float4(0, 0, 0, 0);
            return true;
        case P_DISC:
            rec[0] = make_float4(d[0], d[1], d[2], T); rec[1] = make_float4(d[4], d[5], d[6], d[3]); rec[2] = make_float4(0, 0, 0, 0);
            return true;
        case P_TRIANGLE:
            rec[0] = make_float4(d[0], d[1], d[2], T);
            rec[1] = make_float4(d[4] - d[0], d[5] - d[1], d[6] - d[2], 0);   // edge1 = v1 - v0
            rec[2] = make_float4(d[8] - d[0], d[9] - d[1], d[10] - d[2], 0);  // edge2 = v2 - v0
            return true;
        case P_CONE:
            rec[0] = make_float4(d[0], d[1], d[2], T);
            rec[1] = make_float4(d[8], d[9], d[10], d[11]);
            rec[2] = make_float4(d[3], d[12], d[13], d[14]);
            return true;
        }
        return false;
    }

    /// Result of converting one canonical node: its box and the ref its parent stores for it.
    struct Child {
        float bmin[3], bmax[3];
        uint32_t ref;
    };

    /// A box that is inverted (min > max on some axis: e.g. a sphere with a negative radius, or the empty scene) or holds
    /// a NaN is irregular: the device's fast box test (med3) assumes min <= max, while the reference's comparisons can
    /// still hit such a box through the two planes of its one irregular axis. An infinite bound is irregular too: a plane at
    /// +-inf gives the reference a legitimate candidate with parameter +inf ("intersected", entry stays 1e19), which the
    /// fast test cannot tell from no candidate. Boxes are uploaded as they are; a tree with an irregular box anywhere makes
    /// every box test take the comparison form (Scene::exact_boxes, device_scene.h). With finite, ordered bounds the two
    /// forms agree for every ray, NaN / infinite origins and directions included.
    bool irregular = false;
    void note(const Child &c) {
        for (int k = 0; k < 3; k++)
            irregular = irregular || !(c.bmin[k] <= c.bmax[k]) || std::isinf(c.bmin[k]) || std::isinf(c.bmax[k]);  // NaN fails <=
    }

    /// May the fast kernels visit this tree's nodes in their own order (nearer child first, device_scene.h GD_NEAREST)? Their
    /// certificate rests on boxes that bound what they hold: every child box inside its parent's, every primitive inside its
    /// leaf's box by the very formulas the reference's constructors use (reference src/core.cpp:36-65,80-90,136-150,217-225) —
    /// which only bound a primitive with radii >= 0 and, for a cone, with derived constants (axis, length, width coefficient)
    /// that agree with its two centres and radii. Hostile input (negative radii, NaN, a hand-made tree) fails this and keeps
    /// the reference's order throughout, like a tree with irregular boxes.
    bool disorderly = false;
    /// Some box plane lies within 2^-60 of zero without being zero (a subnormal number, which the device reads as zero, included): the
    /// quick box answers (box_quick.h: gq_plane_ok) are sized for planes that are zero or at least 2^-60 in magnitude — there the
    /// signs of a ray's plane parameters decide the inside test exactly —, such a tree runs its six face tests every time.
    bool subnormal = false;
    static bool tiny(const Child &c) {
        bool r = false;
        for (int k = 0; k < 3; k++) r = r || !gq_plane_ok(c.bmin[k]) || !gq_plane_ok(c.bmax[k]);
        return r;
    }
    static bool box_in_box(const float *cmin, const float *cmax, const float *bmin, const float *bmax) {
        bool ok = true;
        for (int k = 0; k < 3; k++) ok = ok && cmin[k] >= bmin[k] && cmax[k] <= bmax[k] && cmin[k] <= cmax[k];  // NaN fails
        return ok;
    }
    static bool prim_in_box(uint32_t type, const float *d, const float *bmin, const float *bmax) {
        float lo[3], hi[3];
        // Coordinates beyond 2^20 (the hostile classes use 1e10 ... 1e30): the intersectors' own arithmetic cancels there — a
        // vertex at -1e30 swallows the ray origin in `origin - v0` — and a hit parameter that is off by more than the band says
        // nothing about the box it came from. (ulp(2^20) = 0.06: no scene a float path tracer renders sensibly is excluded.)
        static const int LEN[4] = {4, 8, 12, 15};  // (a cone's derived constants 12..14 — width coefficient, cosB, dotAxC1 — included; pads excluded)
        for (int k = 0; k < LEN[type & 3]; k++)
            if (!(std::fabs(d[k]) <= 1048576.0f) && !(type == P_TRIANGLE && (k & 3) == 3)) return false;
        switch (type) {
        case P_SPHERE:
        case P_DISC:  // (the reference bounds a disc by the sphere of its radius)
            if (!(d[3] >= 0)) return false;
            for (int k = 0; k < 3; k++) { lo[k] = d[k] - d[3]; hi[k] = d[k] + d[3]; }
            break;
        case P_TRIANGLE:
            for (int k = 0; k < 3; k++) { lo[k] = std::min(d[k], std::min(d[4 + k], d[8 + k])); hi[k] = std::max(d[k], std::max(d[4 + k], d[8 + k])); }
            break;
        case P_CONE: {
            const float r1 = d[3], r2 = d[7], len = d[11], wc = d[12];
            if (!(r1 >= 0) || !(r2 >= 0) || !(len >= 0)) return false;
            for (int k = 0; k < 3; k++) { lo[k] = std::min(d[k] - r1, d[4 + k] - r2); hi[k] = std::max(d[k] + r1, d[4 + k] + r2); }
            // the intersector works from (centre 1, radius 1, axis, length, width coefficient): they must describe the same frustum
            const float tol = 1.0e-4f * (len + r1 + r2) + 1.0e-30f;
            for (int k = 0; k < 3; k++) if (!(std::fabs(d[k] + len * d[8 + k] - d[4 + k]) <= tol)) return false;
            if (!(std::fabs(r1 + wc * len - r2) <= tol)) return false;
            // ... and so must the constants that POSITION the surface for cone_hit (device_scene.h; reference shaders/cone.glsl:30-135 with
            // the host's src/core.cpp:191-226): a unit axis, dotAxC1 = axis . centre 1 (the F and H terms), cosB from the radii and the
            // length (the normal only — but a record that lies about one derived constant is hand-made: no benefit of the doubt)
            const float ax2 = d[8] * d[8] + d[9] * d[9] + d[10] * d[10];
            if (len > 0 && !(std::fabs(ax2 - 1.0f) <= 1.0e-4f)) return false;
            const float dot_c1 = d[8] * d[0] + d[9] * d[1] + d[10] * d[2];
            if (!(std::fabs(d[14] - dot_c1) <= 1.0e-4f * (std::fabs(d[0]) + std::fabs(d[1]) + std::fabs(d[2]) + 1.0f))) return false;
            float cosb = 0.0f;
            if (std::fabs(r1 - r2) >= 1.0e-7f && len > 0) {
                const float rr = r1 > r2 ? r1 : r2, h = rr * len / std::fabs(r1 - r2);
                cosb = (r1 > r2 ? rr : -rr) / std::sqrt(h * h + rr * rr);
            }
            if (!(std::fabs(d[13] - cosb) <= 1.0e-3f)) return false;
            break;
        }
        default: return false;
        }
        return box_in_box(lo, hi, bmin, bmax);
    }

    /// What the scan keeps of a canonical node: where it lies, the ref its parent stores for it, and (interior nodes) which
    /// entry of the table its upper child is — its lower child is the next entry (pre-order).
    struct NodeInfo {
        uint32_t addr, ref, hi_index, depth;
    };
    std::vector<NodeInfo> table;
    size_t n_recs = 0, n_prims = 0;

    /// Pass 1 — headers only: validates the subtree at quad address `addr` and numbers its records and primitives, in the
    /// order the device arrays keep them (interior nodes in pre-order, so an interior lower child's record directly follows
    /// its parent's; leaves append their primitives). `end` receives the address just past the subtree. Compile (reference
    /// src/bvh.cpp:161-222) lays subtrees out contiguously in pre-order, so an upper child must start exactly where its
    /// sibling's subtree ends: this is what rules out overlapping (DAG-shaped) inputs, whose conversion would otherwise take
    /// exponential time. Returns the node's entry in `table`.
    bool scan(size_t addr, uint32_t depth, uint32_t &index, size_t &end) {
        if (addr + 3 > nq) { err = "node address out of range"; return false; }
        if (depth > 1024) { err = "tree deeper than 1024 levels"; return false; }
        if (depth > max_depth) max_depth = depth;
        num_nodes++;
        index = (uint32_t)table.size();
        table.push_back(NodeInfo{(uint32_t)addr, 0, 0, depth});
        const float *b = q + 4 * addr;
        uint32_t flags = bits(b[8]);
        if (flags & 0x80000000u) {
            uint32_t n = flags & ~0xE0000000u;
            // (a primitive's byte offset, 48 x index, and a record's, 64 x index, fit 32 bits: the kernels address both as base + 32-bit
            //  offset, device_scene.h GD_ADDR32 — 89 million primitives / 67 million interior nodes; the reference's own format ends at
            //  2^31 floats = 76 million triangles with their leaves)
            if (n_prims >= 0x05555540u) { err = "too many primitives"; return false; }
            const uint32_t first = (uint32_t)n_prims;
            size_t a = addr + 3;
            static const int LEN[4] = {1, 2, 3, 4};
            bool all_tris = n >= 1 && n <= 2;
            for (uint32_t i = 0; i < n; i++) {
                if (a + 1 > nq) { err = "primitive header out of range"; return false; }
                uint32_t type = bits(q[4 * a]);
                if (type > 3) { err = "unknown primitive type"; return false; }
                if (type != P_TRIANGLE) all_tris = false;
                type_mask |= 1u << type;
                if (a + 1 + LEN[type] > nq) { err = "primitive data out of range"; return false; }
                a += 1 + LEN[type];
            }
            n_prims += n ? n : 1;  // an empty leaf (empty scene) keeps one dummy record with count 0
            table[index].ref = GD_REF_LEAF | (all_tris ? GD_REF_TRIS : n >= 1 && n <= 2 ? GD_REF_SMALL : 0u) | (n == 2 ? GD_REF_TWO : 0u) | first;
            end = a;
            return true;
        }
        uint32_t lo = bits(b[9]), hi = bits(b[10]);
        if (lo != addr + 3) { err = "lower child does not follow its parent"; return false; }
        if (hi <= lo || hi >= nq) { err = "upper child address out of range"; return false; }
        if (n_recs >= 0x03fffff0u) { err = "too many nodes"; return false; }
        table[index].ref = (uint32_t)n_recs++;
        uint32_t lo_index, hi_index;
        size_t lo_end = 0;
        if (!scan(lo, depth + 1, lo_index, lo_end)) return false;
        if (hi != lo_end) { err = "upper child does not start where the lower subtree ends"; return false; }
        if (!scan(hi, depth + 1, hi_index, end)) return false;
        table[index].hi_index = hi_index;
        return true;
    }

    Child child_of(const NodeInfo &n) const {
        Child c;
        const float *b = q + 4 * (size_t)n.addr;
        for (int k = 0; k < 3; k++) { c.bmin[k] = b[k]; c.bmax[k] = b[4 + k]; }
        c.ref = n.ref;
        return c;
    }

    /// Pass 2 — the entries [from, to) of the table into the device arrays (disjoint writes: any number of threads).
    /// Returns whether one of the boxes written is irregular.
    bool fill(size_t from, size_t to, bool &loose, bool &sub) {
        bool irr = false;
        auto bad = [](const Child &c) {
            bool r = false;
            for (int k = 0; k < 3; k++) r = r || !(c.bmin[k] <= c.bmax[k]) || std::isinf(c.bmin[k]) || std::isinf(c.bmax[k]);  // NaN fails <=
            return r;
        };
        static const int LEN[4] = {1, 2, 3, 4};
        for (size_t i = from; i < to; i++) {
            const NodeInfo &n = table[i];
            const float *b = q + 4 * (size_t)n.addr;
            if (n.ref & GD_REF_LEAF) {
                const uint32_t cnt = bits(b[8]) & ~0xE0000000u;
                float4 *dst = prims.data() + 3 * (size_t)(n.ref & GD_REF_INDEX);
                size_t a = (size_t)n.addr + 3;
                for (uint32_t k = 0; k < cnt; k++) {
                    const uint32_t type = bits(q[4 * a]);
                    pack_prim(type, q + 4 * (a + 1), dst + 3 * k);
                    loose = loose || !prim_in_box(type, q + 4 * (a + 1), b, b + 4);
                    if (k == 0) dst[0].w = fbits(type | (cnt << 2));  // the first primitive carries the leaf's count
                    a += 1 + LEN[type];
                }
                if (cnt == 0) { dst[0] = make_float4(0, 0, 0, fbits(0)); dst[1] = make_float4(0, 0, 0, 0); dst[2] = make_float4(0, 0, 0, 0); }
            } else {
                const Child L = child_of(table[i + 1]), H = child_of(table[n.hi_index]);
                irr = irr || bad(L) || bad(H);
                sub = sub || tiny(L) || tiny(H);
                loose = loose || !box_in_box(L.bmin, L.bmax, b, b + 4) || !box_in_box(H.bmin, H.bmax, b, b + 4);
                float4 *r = recs.data() + 4 * (size_t)n.ref;
                r[0] = make_float4(L.bmin[0], L.bmin[1], L.bmin[2], fbits(L.ref));
                r[1] = make_float4(L.bmax[0], L.bmax[1], L.bmax[2], fbits(H.ref));
                r[2] = make_float4(H.bmin[0], H.bmin[1], H.bmin[2], fbits(n.depth < top_depth ? 1u : 0u));
                r[3] = make_float4(H.bmax[0], H.bmax[1], H.bmax[2], 0);
            }
        }
        return irr;
    }

    /// The whole conversion: scan, then fill on `threads` threads. `root` receives the root's box and ref.
    bool convert(Child &root, unsigned threads) {
        uint32_t index;
        size_t end = 0;
        table.reserve(nq / 6 + 1);
        if (!scan(0, 0, index, end)) return false;
        recs.resize(4 * n_recs);
        prims.resize(3 * n_prims);
        root = child_of(table[0]);
        note(root);
        const size_t n = table.size();
        const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / 16384));
        std::vector<char> irr(parts, 0), loose(parts, 0), sub(parts, 0);
        subnormal = tiny(root);
        std::vector<std::thread> pool;
        auto work = [&](unsigned k) { bool l = false, t = false; irr[k] = fill(n * k / parts, n * (size_t)(k + 1) / parts, l, t) ? 1 : 0; loose[k] = l ? 1 : 0; sub[k] = t ? 1 : 0; };
        for (unsigned k = 1; k < parts; k++) {
            try { pool.emplace_back(work, k); } catch (const std::system_error &) { work(k); }
        }
        work(0);
        for (auto &t : pool) t.join();
        for (char v : irr) irregular = irregular || v;
        for (char v : loose) disorderly = disorderly || v;
        for (char v : sub) subnormal = subnormal || v;
        std::vector<NodeInfo>().swap(table);
        return true;
    }
};

}  // namespace
