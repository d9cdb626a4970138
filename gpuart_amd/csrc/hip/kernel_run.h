// kernel_run.h — k_run: one persistent kernel per pipeline run of the path tracer (gfx950).
//
// The wavefront pipeline of kernels_pipeline.h moves paths between kernels through global queues: a run is a chain of
// 12 dependent launches (k_gen, 6 x k_trace, 5 x k_shade), each of which lasts as long as its slowest ray, and the
// machine is only full while several runs overlap. k_run keeps the same three stages — generate, BVH query, shade —
// but inside ONE launch, with the queues private to a wavefront and held in LDS:
//
//   * every wave owns a `ready` list (paths with a BVH query to run) and a `shade` list (paths whose closest-hit query
//     has finished). Lanes are not tied to paths: a lane that finishes a query hands its path to the shade list and
//     takes the next entry of the ready list, so all 64 lanes keep traversing (the loop of k_trace);
//   * when 64 paths wait in the shade list the whole wave shades them — path_tracing.glsl:182-233 at full lane
//     utilisation, as k_shade did — and appends the survivors to the ready list; when the ready list runs short the
//     wave generates 64 new paths (one 8x8 pixel tile of one pass, path_tracing.glsl:141-175) from a global cursor;
//   * a path's Sun-shadow query of segment s and its closest-hit query of segment s+1 travel as ONE entry: the lane
//     that takes it runs the shadow query first (first accepted hit, applies the Sun term), then the closest-hit query
//     from the same origin. Entries are therefore conserved (one per live path), which bounds the lists:
//     live paths per wave < 256 (proof at PRODUCE below), ready <= 255, shade <= 127;
//   * waves never talk to each other: no device-side termination protocol, no inter-wave visibility question. A wave
//     ends when the cursor is exhausted and its own lists and lanes are empty.
//
// Path state between stages stays in the slot-indexed global arrays of PathBuffers (written and read by different lanes
// of the same wave: ordered with workgroup-scope fences; one wave = one workgroup = one CU's L1). Results are
// bit-identical to the launch pipeline: each path's arithmetic is the same sequence of the same functions, its colour
// goes to its own pixel of the pass's colour plane, and k_accumulate adds the planes in pass order.
#pragma once
#include "kernels_pipeline.h"

#define RUN_RQ 256                  ///< capacity of a wave's ready list
#define RUN_SQ 128                  ///< capacity of a wave's shade list
#define RUN_SLOT 0x1fffffffu        ///< entry: path slot (pass x pixel slot)
#define RUN_F_SHADOW 0x80000000u    ///< entry: run the Sun-shadow query of the segment just shaded first
#define RUN_F_CLOSEST 0x40000000u   ///< entry: (then) run the closest-hit query of the next segment
#define RUN_F_FRESH 0x20000000u     ///< the path has not been shaded yet: segment 0, colorWeight 1, pathColor 0
#ifndef GD_RUN_WAVES
#define GD_RUN_WAVES 5              ///< waves per SIMD: 7.5 KB of LDS per wave -> 21 waves per CU; <= 96 VGPRs
#endif

namespace {

template <bool COUNT, bool REFWORK, int TYPES>
__global__ void __launch_bounds__(BLOCK, GD_RUN_WAVES)
k_run(Scene sc, Frame f, gpuart_params P, SeedBatch seeds, PathBuffers b, int j, int npaths, float4 *passcolor, uint4 *spill,
      unsigned long long *gcounters, TraceTuning tune, uint32_t *cursor) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    __shared__ uint32_t ready[RUN_RQ], shadeq[RUN_SQ];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    const uint32_t total = b.n_slots * b.batch;  // a multiple of 64: one chunk = 64 consecutive path slots (few pixels x the run's passes)
    const bool no_segments = !(P.maxSegments > 0 && 1.0f > P.minWeight);
    const F3 sun = f3(P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]);
    const unsigned long long below = (1ull << lane_id()) - 1;
    WorkCounters wc = {0, 0, {0, 0, 0, 0}, 0, 0};
    uint32_t segments = 0;

    // wave-uniform bookkeeping
    uint32_t n_ready = 0, n_shade = 0;
    bool first_chunk = true, exhausted = false;
    const uint32_t static_end = gridDim.x * BLOCK;  // the first chunk of every wave is static, the cursor starts behind them

    // per-lane query state
    uint32_t ent = SLOT_INVALID;            // the entry this lane works on (slot | flags), SLOT_INVALID: none
    bool shadow = false;                    // the running query is the entry's Sun-shadow query
    F3 ro = f3(0, 0, 0), rd = f3(1, 0, 0), rdiv = f3(1, 1, 1);
    Trav t; t.state = TRAV_DONE; t.closest = 0; t.hit_prim = GD_NO_PRIM; t.node = 0; t.entry = 0;

    for (;;) {
        bool start = false;  // this lane begins a query in this round
        // ---- RETIRE: lanes whose query has finished --------------------------------------------------------------
        // (n_shade <= 63 here, see PRODUCE, so up to 64 appends fit the shade list)
        {
            const bool done = ent != SLOT_INVALID && t.state == TRAV_DONE;
            const uint32_t s = ent & RUN_SLOT;
            bool to_shade = false;
            if (done) {
                if (shadow) {
                    // path_tracing.glsl:239-245: the Sun term counts if nothing lies towards the Sun
                    const float4 term = b.sun[s];
                    F3 pathColor = xyz(b.pc[s]);
                    if (sun_visible(P, ro, sun, t.hit_prim)) pathColor = pathColor + xyz(term);
                    if (ent & RUN_F_CLOSEST) {
                        b.pc[s] = make_float4(pathColor.x, pathColor.y, pathColor.z, 0);
                        ent &= ~RUN_F_SHADOW;
                        shadow = false;
                        start = true;  // the next segment's closest-hit query, same origin
                    } else {
                        path_commit(f, b, passcolor, s, j, npaths, pathColor);  // the path ended with that segment
                        ent = SLOT_INVALID;
                    }
                } else {
                    b.hit[s] = make_uint2(__float_as_uint(t.closest), t.hit_prim);
                    to_shade = true;
                }
            }
            const unsigned long long m = __ballot(to_shade);
            if (to_shade) {
                shadeq[n_shade + (uint32_t)__popcll(m & below)] = ent & (RUN_SLOT | RUN_F_FRESH);
                ent = SLOT_INVALID;
            }
            n_shade += (uint32_t)__popcll(m);
        }
        const uint32_t need = (uint32_t)__popcll(__ballot(ent == SLOT_INVALID));

        // ---- PRODUCE: shade full batches; generate new paths when the ready list cannot feed the idle lanes ------------
        // Bound on the lists. Every live path of this wave is in exactly one place: a lane, the ready list or the shade
        // list (a path's shadow and closest query share one entry). New paths appear only in `generate`, which runs
        // only while n_ready < need (<= idle lanes) and n_shade < 64: live < busy + idle + 64 = 128 before, < 192 after
        // a full chunk, and the loop stops as soon as n_ready >= need, so n_ready <= 127 after generating. Shading moves
        // paths from the shade list to the ready list or ends them. Hence live <= 255 always: ready <= 255, and a shade
        // batch (<= 64 appends) always fits. The shade list is emptied below 64 here, so RETIRE's <= 64 appends fit 128.
        for (;;) {
            const bool want_rays = n_ready < need;
            const bool can_shade = n_shade >= BLOCK || (n_shade > 0 && want_rays && exhausted);
            if (can_shade) {
                // ---- shade up to 64 paths: path_tracing.glsl:182-233, then the loop header of the next segment
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // hit / pc written by other lanes of this wave
                const uint32_t cnt = min(n_shade, (uint32_t)BLOCK);
                n_shade -= cnt;
                uint32_t out = SLOT_INVALID;
                if ((uint32_t)lane_id() < cnt) {
                    const uint32_t e = shadeq[n_shade + lane_id()];
                    const uint32_t s = e & RUN_SLOT;
                    Ray r; r.o = xyz(b.ray_o[s]); r.d = xyz(b.ray_d[s]);
                    const uint2 h = b.hit[s];
                    F3 cw = f3(1, 1, 1), pathColor = f3(0, 0, 0);
                    int seg = 0;
                    if (!(e & RUN_F_FRESH)) {
                        const float4 c4 = b.cw[s];
                        cw = xyz(c4); seg = (int)__float_as_uint(c4.w);
                        pathColor = xyz(b.pc[s]);
                    }
                    F3 rstart = r.o, rdir = r.d;
                    if (COUNT) segments++;
                    const float4 seed = seeds.seed[slot_pass(b, s)];
                    ShadeResult sr = path_shade(sc, P, seed, seg, r, __uint_as_float(h.x), h.y, rstart, rdir, cw, pathColor);
                    if (sr.broke) {
                        uint32_t lx, ly; F3 rs0, rd0;
                        slot_pixel(f, slot_pixel_slot(b, s), lx, ly);
                        camera_ray(f, f.x0 + lx, frame_y(f, ly), rs0, rd0);
                        path_commit(f, b, passcolor, s, j, npaths, path_finish(P, rd0, seg, sr.ush, sr.specular, pathColor));
                    } else {
                        const bool go_on = sr.next == PATH_CONTINUES;
                        const bool sh = sr.want_shadow && (REFWORK || sr.sun_matters);
                        if (sh) b.sun[s] = make_float4(sr.sun_term.x, sr.sun_term.y, sr.sun_term.z, __uint_as_float(go_on ? 0u : 1u));
                        if (go_on || sh) {
                            b.ray_o[s] = make_float4(rstart.x, rstart.y, rstart.z, 0);
                            b.pc[s] = make_float4(pathColor.x, pathColor.y, pathColor.z, 0);
                            out = s | (sh ? RUN_F_SHADOW : 0u) | (go_on ? RUN_F_CLOSEST : 0u);
                        } else {
                            path_commit(f, b, passcolor, s, j, npaths, pathColor);  // i >= 1: no special case
                        }
                        if (go_on) {
                            b.ray_d[s] = make_float4(rdir.x, rdir.y, rdir.z, 0);
                            b.cw[s] = make_float4(cw.x, cw.y, cw.z, __uint_as_float((uint32_t)(seg + 1)));
                        }
                    }
                }
                const unsigned long long m = __ballot(out != SLOT_INVALID);
                if (out != SLOT_INVALID) ready[n_ready + (uint32_t)__popcll(m & below)] = out;
                n_ready += (uint32_t)__popcll(m);
                continue;
            }
            if (!want_rays || exhausted) break;
            // ---- generate 64 paths: the next 8x8 pixel tile of a pass (path_tracing.glsl:141-175)
            uint32_t base;
            if (first_chunk) {
                first_chunk = false;
                base = blockIdx.x * BLOCK;
            } else {
                base = 0;
                if (lane_id() == 0) base = atomicAdd(cursor, (uint32_t)BLOCK);
                base = __shfl(base, 0, 64) + static_end;
            }
            if (base >= total) { exhausted = true; continue; }
            const uint32_t slot = base + lane_id();
            uint32_t lx, ly;
            uint32_t out = SLOT_INVALID;
            if (slot_pixel(f, slot_pixel_slot(b, slot), lx, ly)) {
                F3 rs0, rd0, rs, rdd;
                camera_ray(f, f.x0 + lx, frame_y(f, ly), rs0, rd0);
                if (no_segments) {  // the GLSL loop body never runs: i == 0 and no user-sphere hit
                    path_commit(f, b, passcolor, slot, j, npaths, path_finish(P, rd0, 0, false, false, f3(0, 0, 0)));
                } else {
                    path_begin(P, seeds.seed[slot_pass(b, slot)], j, rs0, rd0, rs, rdd);
                    b.ray_o[slot] = make_float4(rs.x, rs.y, rs.z, 0);
                    b.ray_d[slot] = make_float4(rdd.x, rdd.y, rdd.z, 0);
                    out = slot | RUN_F_CLOSEST | RUN_F_FRESH;
                }
            }
            const unsigned long long m = __ballot(out != SLOT_INVALID);
            if (out != SLOT_INVALID) ready[n_ready + (uint32_t)__popcll(m & below)] = out;
            n_ready += (uint32_t)__popcll(m);
        }

        // ---- REFILL: idle lanes take the newest entries of the ready list ---------------------------------------------
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // rays / lists written by other lanes of this wave
            const unsigned long long idle = __ballot(ent == SLOT_INVALID);
            const uint32_t take = min((uint32_t)__popcll(idle), n_ready);
            const uint32_t rank = (uint32_t)__popcll(idle & below);
            if (ent == SLOT_INVALID && rank < take) {
                ent = ready[n_ready - 1 - rank];  // newest first (oldest first measured the same)
                shadow = (ent & RUN_F_SHADOW) != 0;
                ro = xyz(b.ray_o[ent & RUN_SLOT]);
                start = true;
            }
            n_ready -= take;
        }
        if (start) {
            rd = shadow ? sun : xyz(b.ray_d[ent & RUN_SLOT]);
            rdiv = f3(1 / rd.x, 1 / rd.y, 1 / rd.z);
            trav_init(sc, Ray{ro, rd}, rdiv, t, st, &wc, COUNT);
        }
        if (__ballot(ent != SLOT_INVALID) == 0) break;  // nothing in flight; PRODUCE could not make anything: all done

        // ---- TRAVERSE until enough lanes have finished (a lane without an entry is in state DONE) ---------------------
        for (;;) {
            if (t.state == TRAV_DESCEND) trav_step_box<COUNT>(sc, Ray{ro, rd}, rdiv, t, st, COUNT ? &wc : nullptr);
            unsigned long long at_leaf = __ballot((t.state & 1) != 0);  // TRAV_LEAF = 1, TRAV_LEAF_TRIS = 3
            unsigned long long descending = __ballot(t.state == TRAV_DESCEND);
            const uint32_t waiting = (uint32_t)__popcll(at_leaf);
            if (at_leaf && (waiting >= tune.leaf_lanes || tune.leaf_share * waiting >= waiting + (uint32_t)__popcll(descending))) {
                if (t.state & 1) {
                    trav_step_leaf<false, COUNT, TYPES>(sc, Ray{ro, rd}, t, st, COUNT ? &wc : nullptr);
                    // the reference only asks a shadow query whether anything was hit: one accepted hit settles it
                    if (!REFWORK && shadow && t.hit_prim != GD_NO_PRIM) t.state = TRAV_DONE;
                }
                descending = __ballot(t.state == TRAV_DESCEND);
                at_leaf = __ballot((t.state & 1) != 0);
            }
            const unsigned long long busy = descending | at_leaf;
            if (!busy) break;
            if (64u - (uint32_t)__popcll(busy) >= tune.refill_lanes) {
                // worth a round of RETIRE / PRODUCE / REFILL if that can put idle lanes back to work: a finished lane
                // holds a path that goes on, or the lists / the cursor still have something
                const unsigned long long finished = __ballot(ent != SLOT_INVALID && t.state == TRAV_DONE);
                if (finished || n_ready || n_shade || !exhausted) break;
            }
        }
    }
    if (COUNT) flush_counters(wc, segments, gcounters);
}

}  // namespace
