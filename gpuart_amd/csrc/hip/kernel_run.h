// kernel_run.h — k_run: one persistent kernel per pipeline run of the path tracer (gfx950).
//
// The wavefront pipeline of kernels_pipeline.h moves paths between kernels through global queues: a run is a chain of
// 12 dependent launches (k_gen, 6 x k_trace, 5 x k_shade), each of which lasts as long as its slowest ray, and the
// machine is only full while several runs overlap. k_run keeps the same three stages — generate, BVH query, shade —
// but inside ONE launch, with the queues private to a wavefront and held in LDS:
//
//   * every wave owns a `ready` list (paths with a BVH query to run) and a `shade` list (paths whose closest-hit query
//     has finished). Lanes are not tied to paths: a lane that finishes a query hands its path to the shade list and
//     takes the OLDEST entry of the ready list (a ring), so all 64 lanes keep traversing (the loop of k_trace);
//   * when 64 paths wait in the shade list the whole wave shades them — path_tracing.glsl:182-233 at full lane
//     utilisation, as k_shade did — and appends the survivors to the ready list; when the ready list runs short the
//     wave generates 64 new paths (one 8x8 pixel tile of one pass, path_tracing.glsl:141-175) from a global cursor;
//   * a path's Sun-shadow query of segment s and its closest-hit query of segment s+1 are two entries that two lanes run at
//     the same time (a path is a chain of up to 5 + 5 queries; run one after the other that chain, not the machine, sets the
//     time of a pass observed alone: 2.9 ms, against 2.2 now). The path is shaded again when BOTH have finished: the two
//     entries carry a join flag, a per-slot counter in global memory is 2, whoever decrements it to 0 hands the path to the
//     shade list. Live paths per wave are counted (n_live < 192, proof at PRODUCE below): ready <= 382 entries, shade <= 127;
//   * what a single pass costs beyond its work is the tail (tools/run_timeline.py): the cursor runs dry after 0.85 ms with the
//     machine full, and the paths born last still have a chain of up to five segments ahead, ~0.25 ms each while the machine
//     is loaded; busy lanes then halve every ~0.25 ms. Hence: shade as soon as lanes would idle (RUN_SHADE_MIN), do not stop
//     the traversing lanes for every single finished one (RUN_TAIL_DIV), never generate ahead of need (RUN_LIVE_MAX);
//   * waves never talk to each other: no device-side termination protocol, no inter-wave visibility question. A wave
//     ends when the cursor is exhausted and its own lists and lanes are empty.
//
// Path state between stages stays in the slot-indexed global arrays of PathBuffers (written and read by different lanes
// of the same wave: ordered with workgroup-scope fences; one wave = one workgroup = one CU's L1). Results are
// bit-identical to the launch pipeline: each path's arithmetic is the same sequence of the same functions, its colour
// goes to its own pixel of the pass's colour plane, and k_accumulate adds the planes in pass order.
#pragma once
#include <type_traits>

#include "kernels_pipeline.h"

#ifndef RUN_RQ
#define RUN_RQ 512                  ///< capacity of a wave's ready list (at most two entries per live path)
#endif
#define RUN_SQ 128                  ///< capacity of a wave's shade list
#define RUN_SLOT 0x0fffffffu        ///< entry: path slot (pass x pixel slot)
#define RUN_F_REWALK 0x10000000u    ///< (lane only) the query is on its second walk, in the reference's order (device_scene.h trav_settle)
#define RUN_F_SHADOW 0x80000000u    ///< entry: the Sun-shadow query of the segment just shaded (else: the closest-hit query of the next one)
#define RUN_F_JOIN 0x40000000u      ///< the path's other query is under way too: the last of the two to finish hands the path on
#define RUN_F_FRESH 0x20000000u     ///< the path has not been shaded yet: segment 0, colorWeight 1, pathColor 0
#ifndef RUN_SHADE_MIN
#define RUN_SHADE_MIN 32              ///< idle lanes and an empty ready list: shade this many waiting paths rather than generate new ones
#endif
#ifndef RUN_LIVE_MAX
#define RUN_LIVE_MAX 0                ///< generate ahead while the wave holds at most this many live paths (0: only when lanes would idle).
                                      ///< Measured (gpurun_out/ab_eager.txt): 128 / 192 / 256 -> 2.19 / 2.40 / 2.68 ms against 2.16 for one pass alone:
                                      ///< paths waiting in a list do not progress, and the later the cursor runs dry the better it balances the waves
#endif
#ifndef RUN_TAIL_DIV
#define RUN_TAIL_DIV 2                ///< tail: retire / shade once the finished lanes are 1/RUN_TAIL_DIV of the traversing ones
#endif
#ifndef RUN_PIPE
#define RUN_PIPE 1                    ///< thin modes fetch ahead: what a ray's next step needs is requested as soon as the step before has decided
                                      ///< it, and a leaf entered in this round is tested in the next one, after its triangles have arrived
#endif
#ifndef RUN_THIN
#define RUN_THIN 4                    ///< most lanes per ray in the tail (1: never leave wide mode; 2; 4). Once the cursor is dry and a wave is
                                      ///< down to 32 (16) live paths and queries in flight, its rays are carried by pairs (quads) of lanes
                                      ///< (device_scene.h "thin-wave modes"): a draining wave otherwise pays whole instructions for a few lanes
#endif
// Occupancy (gpurun_out/ab_occ.txt): 5 waves per SIMD need <= 96 VGPRs = 9 spilled registers and RUN_RQ 256 (7.5 KB of LDS): 64 passes as
// one run 1.08 against 1.175 ms per pass, but one pass alone 2.37 against 2.21 and a 1/8 share with 20 passes unchanged; 6 waves (80 VGPRs,
// 28 spills, 6-entry ring) lose everywhere. The shading code inside the kernel sets the register count; 4 waves it stays.
#ifndef GD_RUN_WAVES
#define GD_RUN_WAVES 4              ///< waves per SIMD: 8.5 KB of LDS per wave -> 18 waves per CU; <= 128 VGPRs (shading code inside)
#endif

#ifdef GD_RUN_TIMELINE
// diagnostic build (tools/run_timeline.py): per wave, the 100 MHz clock at its start, when the cursor ran dry, at its end,
// and the lane-rounds it spent traversing (active lanes summed over the rounds of the TRAVERSE loop / rounds)
__device__ unsigned long long g_run_timeline[24 * 8192];
__device__ unsigned long long g_run_hist[2 * 128];  // busy lane-time and wave-time per 25 us bucket
#endif

#ifdef GD_RUN_TIMELINE
// one round of the TRAVERSE loop with `n` busy lanes (replicas included)
#define GD_RUN_TL_ROUND(n)                                                                                                                     \
    {                                                                                                                                          \
        tl_lanes += (unsigned long long)(n); tl_rounds++; if (exhausted) tl_tail_rounds++;                                                     \
        if (M == 2) tl_r2++; else if (M == 4) tl_r4++;                                                                                         \
        if (!exhausted) { tl_nready += n_ready; tl_nshade += n_shade; }                                                                        \
        const unsigned long long now = wall_clock64();                                                                                         \
        const unsigned bucket = (unsigned)min((now - (tl_zero ? tl_zero : tl_start)) / 2500ull, 127ull);                                       \
        if (lane_id() == 0) { tl_hist[bucket] += (unsigned)(now - tl_prev) * (unsigned)(n); tl_hist[128 + bucket] += (unsigned)(now - tl_prev); } \
        tl_prev = now;                                                                                                                         \
    }
#else
#define GD_RUN_TL_ROUND(n)
#endif

namespace {

static_assert((RUN_RQ & (RUN_RQ - 1)) == 0 && RUN_RQ >= 256, "the ready list is a ring indexed modulo RUN_RQ and holds two entries for each of RUN_RQ / 2 live paths (generation stops there)");
static_assert(RUN_SQ >= 2 * BLOCK, "RETIRE appends up to 64 entries to a shade list of up to 63");

template <bool COUNT, bool REFWORK, int TYPES>
__global__ void __launch_bounds__(BLOCK, GD_RUN_WAVES)
k_run(Scene sc, Frame f, gpuart_params P, SeedBatch seeds, PathBuffers b, int j, int npaths, float4 *passcolor, uint4 *spill,
      unsigned long long *gcounters, TraceTuning tune, uint32_t *cursor) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    __shared__ uint32_t warm_sink[BLOCK];  // where the thin-wave steps' L1-warming loads land (device_scene.h ThinPrefetch); never read
    __shared__ uint32_t ready[RUN_RQ], shadeq[RUN_SQ];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    const uint32_t total = b.n_slots * b.batch;  // a multiple of 64: one chunk = 64 consecutive path slots (few pixels x the run's passes)
    const bool no_segments = !(P.maxSegments > 0 && 1.0f > P.minWeight);
    const F3 sun = f3(P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]);
    WorkCounters wc = {0, 0, {0, 0, 0, 0}, 0, 0, 0};
    uint32_t segments = 0;

    // wave-uniform bookkeeping
    uint32_t n_ready = 0, n_shade = 0, r_head = 0, n_live = 0;  // the ready list is a ring: oldest entry at r_head
    bool first_chunk = true, exhausted = false;
    const uint32_t static_end = gridDim.x * BLOCK;  // the first chunk of every wave is static, the cursor starts behind them

    // Thin-wave modes (tail only): M lanes per ray, rays at lanes [M q, M q + M) as identical replicas; `lead` = the first lane of
    // every group, `sub` = this lane's index in its group. Never entered by the
    // counting variants (their counters are per lane) and by trees with irregular boxes (comparison-form box tests).
    constexpr bool THIN_OK = RUN_THIN > 1 && !COUNT && GD_BOXES_OF(TYPES) == GD_BOXES_FAST;
    // closest-hit queries enter the nearer child first (device_scene.h, GD_NEAREST); the counters of mode 4 count that walk
    constexpr bool NEAR = GD_NEAREST_OF(TYPES) && !REFWORK;
    uint32_t M = 1, sub = 0;                                   // M wave-uniform
    unsigned long long lead = ~0ull;                           // wave-uniform
    __shared__ uint32_t xfer[BLOCK];

    // per-lane query state
    uint32_t ent = SLOT_INVALID;            // the entry this lane works on (slot | flags), SLOT_INVALID: none
    F3 ro = f3(0, 0, 0), rd = f3(1, 0, 0), rdiv = f3(1, 1, 1);
    Trav t; t.state = TRAV_DONE; t.closest = 0; t.hit_prim = GD_NO_PRIM; t.node = 0; t.entry = 0; t.second = 0;

#ifdef GD_RUN_TIMELINE
    unsigned long long tl_start = wall_clock64(), tl_dry = 0, tl_lanes = 0, tl_rounds = 0, tl_prev = tl_start, tl_nready = 0, tl_nshade = 0;
    unsigned long long tl_tail_trav = 0, tl_tail_other = 0, tl_tail_rounds = 0, tl_tail_trig = 0, tl_mark = tl_start;
    unsigned long long tl_m2 = 0, tl_m4 = 0, tl_r2 = 0, tl_r4 = 0;  // when the wave went to pairs / quads, rounds in either mode
    unsigned long long tp_box = 0, tp_nbox = 0, tp_leaf = 0, tp_nleaf = 0, tp_loop = 0, tp_busy = 0;  // shader-clock cycles of the quad rounds: box steps, leaf steps, whole rounds
    __shared__ unsigned tl_hist[2 * 128];
    tl_hist[lane_id()] = 0; tl_hist[64 + lane_id()] = 0; tl_hist[128 + lane_id()] = 0; tl_hist[192 + lane_id()] = 0;
    const unsigned long long tl_zero = g_run_hist[2 * 128 - 1];  // the host stores the launch's reference clock there (0: use own start)
#endif
    for (;;) {
        bool start = false;  // this lane begins a query in this round
        // ---- a finished nearest-first query that cannot vouch for its answer walks again, in the reference's order (every replica alike)
        if (NEAR && ent != SLOT_INVALID && t.state == TRAV_DONE && trav_settle<NEAR>(t, !(ent & (RUN_F_SHADOW | RUN_F_REWALK)))) {
            ent |= RUN_F_REWALK;
            if (COUNT) wc.rewalks++;
            trav_init<GD_BOXES_OF(TYPES)>(sc, Ray{ro, rd}, rdiv, t, st, &wc, false);
        }
        // ---- RETIRE: lanes whose query has finished --------------------------------------------------------------
        // (n_shade <= 63 here, see PRODUCE, so up to 64 appends fit the shade list)
        {
            const bool done = ent != SLOT_INVALID && t.state == TRAV_DONE && (!THIN_OK || sub == 0);  // a ray's first replica retires it
            const uint32_t s = ent & RUN_SLOT;
            bool to_shade = false, joins = false, ended = false;
            if (done) {
                if (ent & RUN_F_SHADOW) {
                    // path_tracing.glsl:239-245: the Sun term counts if nothing lies towards the Sun
                    const float4 term = b.sun[s];
                    F3 pathColor = xyz(b.pc[s]);
                    if (sun_visible(P, ro, sun, t.hit_prim)) pathColor = pathColor + xyz(term);
                    if (__float_as_uint(term.w) & 1u) {
                        path_commit(f, b, passcolor, s, j, npaths, pathColor);  // the path ended with that segment
                        ended = true;
                    } else {
                        b.pc[s] = make_float4(pathColor.x, pathColor.y, pathColor.z, 0);
                        joins = true;  // its next closest-hit query is under way (or already back)
                    }
                } else {
                    b.hit[s] = make_uint2(__float_as_uint(t.closest), t.hit_prim);
                    joins = (ent & RUN_F_JOIN) != 0;
                    to_shade = !joins;
                }
            }
            if (__ballot(joins)) {
                // what this lane stored must be in memory before the counter says so (both parties are lanes of this wave)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (joins) to_shade = atomicSub(&b.queue[0][s], 1u) == 1u;
            }
            const unsigned long long m = __ballot(to_shade);
            if (to_shade) shadeq[n_shade + rank_below(m)] = ent & (RUN_SLOT | RUN_F_FRESH);
            n_shade += (uint32_t)__popcll(m);
            n_live -= (uint32_t)__popcll(__ballot(ended));
            if (ent != SLOT_INVALID && t.state == TRAV_DONE) ent = SLOT_INVALID;
        }
        const uint32_t need = (uint32_t)__popcll(__ballot(ent == SLOT_INVALID) & lead);

        // ---- PRODUCE: shade full batches; generate new paths when the ready list cannot feed the idle lanes ------------
        // Bound on the lists. n_live counts the wave's live paths: +1 per generated path, -1 where a path is committed. A live
        // path is represented by one or two entries in lanes / the ready list, or by one entry of the shade list. New paths
        // appear only in `generate`, which (RUN_LIVE_MAX = 0) runs only while n_ready < need (<= idle lanes) and n_shade < 64
        // (or < RUN_SHADE_MIN with lanes idle): fewer than 64 paths in lanes, fewer than 64 in the ready list, fewer than 64 in
        // the shade list, i.e. n_live < 192 before and < 256 after a chunk; whatever RUN_LIVE_MAX says, generation stops at
        // n_live + 64 > 256. Hence ready <= 2 x 256 = 512 entries = RUN_RQ always, and a shade batch (<= 128 appends) fits.
        // The shade list is emptied below 64 here, so RETIRE's <= 64 appends fit 128.
        for (;;) {
            const bool want_rays = n_ready < need;
            const bool can_shade = n_shade >= BLOCK || (n_shade > 0 && want_rays && (exhausted || n_shade >= RUN_SHADE_MIN));
            if (can_shade) {
                // ---- shade up to 64 paths: path_tracing.glsl:182-233, then the loop header of the next segment
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // hit / pc written by other lanes of this wave
                const uint32_t cnt = min(n_shade, (uint32_t)BLOCK);
                n_shade -= cnt;
                uint32_t out = SLOT_INVALID, out_sh = SLOT_INVALID;
                bool ended = false;
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
                    tile_cost_add(f, b, s);
                    const float4 seed = seeds.seed[slot_pass(b, s)];
                    ShadeResult sr = path_shade(sc, P, seed, seg, r, __uint_as_float(h.x), h.y, rstart, rdir, cw, pathColor);
                    if (sr.broke) {
                        uint32_t lx, ly; F3 rs0, rd0;
                        slot_pixel(f, slot_pixel_slot(b, s), lx, ly);
                        camera_ray(f, f.x0 + lx, frame_y(f, ly), rs0, rd0);
                        path_commit(f, b, passcolor, s, j, npaths, path_finish(P, rd0, seg, sr.ush, sr.specular, pathColor));
                        ended = true;
                    } else {
                        const bool go_on = sr.next == PATH_CONTINUES;
                        const bool sh = sr.want_shadow && (REFWORK || sr.sun_matters);
                        if (sh) b.sun[s] = make_float4(sr.sun_term.x, sr.sun_term.y, sr.sun_term.z, __uint_as_float(go_on ? 0u : 1u));
                        if (go_on || sh) {
                            b.ray_o[s] = make_float4(rstart.x, rstart.y, rstart.z, 0);
                            b.pc[s] = make_float4(pathColor.x, pathColor.y, pathColor.z, 0);
                            const uint32_t join = go_on && sh ? RUN_F_JOIN : 0u;
                            if (join) b.queue[0][s] = 2u;  // the path is shaded again when both queries are back
                            if (go_on) out = s | join;
                            if (sh) out_sh = s | RUN_F_SHADOW | join;
                        } else {
                            path_commit(f, b, passcolor, s, j, npaths, pathColor);  // i >= 1: no special case
                            ended = true;
                        }
                        if (go_on) {
                            b.ray_d[s] = make_float4(rdir.x, rdir.y, rdir.z, 0);
                            b.cw[s] = make_float4(cw.x, cw.y, cw.z, __uint_as_float((uint32_t)(seg + 1)));
                        }
                    }
                }
                // oldest first: the short shadow queries of a batch go before its closest-hit queries
                n_live -= (uint32_t)__popcll(__ballot(ended));
                unsigned long long m = __ballot(out_sh != SLOT_INVALID);
                if (out_sh != SLOT_INVALID) ready[(r_head + n_ready + rank_below(m)) & (RUN_RQ - 1)] = out_sh;
                n_ready += (uint32_t)__popcll(m);
                m = __ballot(out != SLOT_INVALID);
                if (out != SLOT_INVALID) ready[(r_head + n_ready + rank_below(m)) & (RUN_RQ - 1)] = out;
                n_ready += (uint32_t)__popcll(m);
                continue;
            }
            // ahead of need while the wave holds few live paths (RUN_LIVE_MAX): the sooner a path is born the sooner its chain of
            // up to five segments ends; never beyond 256 live paths (<= 512 entries)
            if (exhausted || n_live + BLOCK > (uint32_t)(RUN_RQ / 2) || !(want_rays || n_live + BLOCK <= (uint32_t)RUN_LIVE_MAX)) break;
            // ---- generate 64 paths: the next 8x8 pixel tile of a pass (path_tracing.glsl:141-175)
            uint32_t base;
            if (first_chunk) {
                first_chunk = false;
                base = blockIdx.x * BLOCK;
            } else {
                base = 0;
                if (lane_id() == 0) base = atomicAdd(cursor, (uint32_t)BLOCK);
                base = wave_value(base) + static_end;
            }
            if (base >= total) {
                exhausted = true;
#ifdef GD_RUN_TIMELINE
                tl_dry = wall_clock64();
#endif
                continue;
            }
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
                    out = slot | RUN_F_FRESH;
                }
            }
            const unsigned long long m = __ballot(out != SLOT_INVALID);
            if (out != SLOT_INVALID) ready[(r_head + n_ready + rank_below(m)) & (RUN_RQ - 1)] = out;
            n_ready += (uint32_t)__popcll(m);
            n_live += (uint32_t)__popcll(m);
        }

        // ---- THIN: the cursor is dry and few paths are left: fewer ray positions, more lanes per ray ---------------------------
        if (THIN_OK && exhausted && M < (uint32_t)RUN_THIN) {
            const unsigned long long flying = __ballot(ent != SLOT_INVALID) & lead;
            const uint32_t most = max(n_live, (uint32_t)__popcll(flying));  // a path has up to two queries, the surplus waits in the ready list
            const uint32_t to = most <= BLOCK / 4 && RUN_THIN >= 4 ? 4u : most <= BLOCK / 2 ? 2u : 1u;
            if (to > M) {
                uint32_t unused = 0;
                thin_regroup(to, flying, xfer, ring_a, ring_b, spill, ent, unused, ro, rd, rdiv, t, st);
                if ((uint32_t)lane_id() / to >= (uint32_t)__popcll(flying)) ent = SLOT_INVALID;  // the groups beyond the rays in flight are idle
                M = to;
#ifdef GD_RUN_TIMELINE
                if (M == 2) tl_m2 = wall_clock64(); else { tl_m4 = wall_clock64(); }
#endif
                sub = (uint32_t)lane_id() & (M - 1);
                lead = M == 4 ? 0x1111111111111111ull : 0x5555555555555555ull;
            }
        }

        // ---- REFILL: idle lanes take the oldest entries of the ready list ---------------------------------------------
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // rays / lists written by other lanes of this wave
            const unsigned long long idle = __ballot(ent == SLOT_INVALID) & lead;
            const uint32_t take = min((uint32_t)__popcll(idle), n_ready);
            // `idle` holds the first lanes of the idle groups: a replica further back in such a group counts its own first lane too
            const uint32_t rank = rank_below(idle) - (sub != 0 ? 1u : 0u);  // the same for every replica of a group
            if (ent == SLOT_INVALID && rank < take) {
                ent = ready[(r_head + rank) & (RUN_RQ - 1)];  // oldest first: with paths generated ahead, the youngest segments go first
                ro = xyz(b.ray_o[ent & RUN_SLOT]);
                start = true;
            }
            n_ready -= take;
            r_head = (r_head + take) & (RUN_RQ - 1);
        }
        if (start) {
            rd = (ent & RUN_F_SHADOW) ? sun : xyz(b.ray_d[ent & RUN_SLOT]);
            rdiv = f3(1 / rd.x, 1 / rd.y, 1 / rd.z);
            trav_init<GD_BOXES_OF(TYPES)>(sc, Ray{ro, rd}, rdiv, t, st, &wc, COUNT, NEAR && !(ent & RUN_F_SHADOW));
        }
        if (__ballot(ent != SLOT_INVALID) == 0) break;  // nothing in flight; PRODUCE could not make anything: all done

        // ---- TRAVERSE until enough lanes have finished (a lane without an entry is in state DONE) ---------------------
#ifdef GD_RUN_TIMELINE
        { const unsigned long long now = wall_clock64(); if (exhausted) { tl_tail_other += now - tl_mark; tl_tail_trig++; } tl_mark = now; }
#endif
        if (THIN_OK && M > 1) {
            // the same loop with M lanes per ray: ballots count groups (their first lanes), thresholds are in lanes
            auto thin_rounds = [&](auto width) {
                constexpr int W = decltype(width)::value;
                constexpr unsigned long long LEAD = W == 4 ? 0x1111111111111111ull : 0x5555555555555555ull;
#if RUN_PIPE
                // Software-pipelined: a ray's record / triangles are requested the moment the step before has decided what comes next
                // (thin_fetch), and a triangle leaf entered in this round's box phase is tested in the NEXT round's leaf phase — its
                // data then arrives beside the other rays' steps instead of stalling the whole wave at the head of the leaf phase.
                ThinFetch pf;
                pf.a = pf.b = pf.c = make_float4(0, 0, 0, 0);
                thin_fetch<W>(sc, t, sub, pf);
                bool fresh = false;  // this lane's leaf was entered in this round: its triangles are on their way
#endif
                for (;;) {
#ifdef GD_RUN_TIMELINE
                    const unsigned long long pc0 = __builtin_amdgcn_s_memtime();
#endif
#if RUN_PIPE
                    if (t.state == TRAV_DESCEND) {
                        trav_step_box_thin_on<W, NEAR>(ro, rd, rdiv, t, st, sub, pf, sc.box_slack, !(ent & (RUN_F_SHADOW | RUN_F_REWALK)), ThinPrefetch{sc.recs, sc.prims, warm_sink});
                        thin_fetch<W>(sc, t, sub, pf);
                        fresh = (t.state & 8) != 0;
                    }
                    const bool leaf_now = (t.state & 1) != 0 && !fresh;
#else
                    if (t.state == TRAV_DESCEND) trav_step_box_thin<W, NEAR>(sc, ro, rd, rdiv, t, st, sub, !(ent & (RUN_F_SHADOW | RUN_F_REWALK)), warm_sink);
                    const bool leaf_now = (t.state & 1) != 0;
#endif
#ifdef GD_RUN_TIMELINE
                    if (W == 4) { tp_box += __builtin_amdgcn_s_memtime() - pc0; tp_nbox++; }
#endif
                    unsigned long long at_leaf = __ballot(leaf_now) & LEAD;
                    unsigned long long busy = __ballot(t.state != TRAV_DONE) & LEAD;
                    const uint32_t waiting = (uint32_t)__popcll(at_leaf);
                    if (at_leaf && (W * waiting >= tune.leaf_lanes || tune.leaf_share * waiting >= (uint32_t)__popcll(busy))) {
#ifdef GD_RUN_TIMELINE
                        const unsigned long long pl0 = __builtin_amdgcn_s_memtime();
#endif
                        if (leaf_now) {
#if RUN_PIPE
                            trav_step_leaf_thin_on<W, TYPES, NEAR>(sc, ro, rd, t, st, sub, pf, !(ent & (RUN_F_SHADOW | RUN_F_REWALK)));
#else
                            trav_step_leaf_thin<W, TYPES, NEAR>(sc, ro, rd, t, st, sub, !(ent & (RUN_F_SHADOW | RUN_F_REWALK)));
#endif
                            if (!REFWORK && (ent & RUN_F_SHADOW) && t.hit_prim != GD_NO_PRIM) t.state = TRAV_DONE;
#if RUN_PIPE
                            thin_fetch<W>(sc, t, sub, pf);
#endif
                        }
                        busy = __ballot(t.state != TRAV_DONE) & LEAD;
#ifdef GD_RUN_TIMELINE
                        if (W == 4) { tp_leaf += __builtin_amdgcn_s_memtime() - pl0; tp_nleaf++; }
#endif
                    }
#if RUN_PIPE
                    fresh = false;
#endif
#ifdef GD_RUN_TIMELINE
                    if (W == 4) { tp_loop += __builtin_amdgcn_s_memtime() - pc0; tp_busy += (uint32_t)__popcll(busy); }
#endif
                    GD_RUN_TL_ROUND((uint32_t)__popcll(busy))
                    if (!busy) break;
                    if ((uint32_t)BLOCK - W * (uint32_t)__popcll(busy) >= tune.refill_lanes) {
                        const uint32_t finished = (uint32_t)__popcll(__ballot(ent != SLOT_INVALID && t.state == TRAV_DONE) & LEAD);
                        if (n_ready || n_shade || finished * RUN_TAIL_DIV >= (uint32_t)__popcll(busy)) break;  // (the cursor is dry)
                    }
                }
            };
            if (M == 2) thin_rounds(std::integral_constant<int, 2>());
            else thin_rounds(std::integral_constant<int, 4>());
        } else
        for (;;) {
            if (t.state == TRAV_DESCEND) trav_step_box<COUNT, GD_BOXES_OF(TYPES), NEAR>(sc, Ray{ro, rd}, rdiv, t, st, COUNT ? &wc : nullptr, !(ent & (RUN_F_SHADOW | RUN_F_REWALK)));
            unsigned long long at_leaf = __ballot((t.state & 1) != 0);  // the leaf states are the odd ones
            unsigned long long descending = __ballot(t.state == TRAV_DESCEND);
            const uint32_t waiting = (uint32_t)__popcll(at_leaf);
            if (at_leaf && (waiting >= tune.leaf_lanes || tune.leaf_share * waiting >= waiting + (uint32_t)__popcll(descending))) {
                if (t.state & 1) {
                    trav_step_leaf<false, COUNT, TYPES, NEAR>(sc, Ray{ro, rd}, t, st, COUNT ? &wc : nullptr, !(ent & (RUN_F_SHADOW | RUN_F_REWALK)));
                    // the reference only asks a shadow query whether anything was hit: one accepted hit settles it
                    if (!REFWORK && (ent & RUN_F_SHADOW) && t.hit_prim != GD_NO_PRIM) t.state = TRAV_DONE;
                }
                descending = __ballot(t.state == TRAV_DESCEND);
                at_leaf = __ballot((t.state & 1) != 0);
            }
            const unsigned long long busy = descending | at_leaf;
            GD_RUN_TL_ROUND((uint32_t)__popcll(busy))
            if (!busy) break;
            if (64u - (uint32_t)__popcll(busy) >= tune.refill_lanes) {
                // worth a round of RETIRE / PRODUCE / REFILL if that can put idle lanes back to work: a finished lane
                // holds a path that goes on, or the lists / the cursor still have something
                // In the tail (cursor dry, lists empty) only finished lanes can feed the idle ones, through a shade step during which
                // the lanes still traversing stand still: wait until the finished lanes are a fair share of those.
                const uint32_t finished = (uint32_t)__popcll(__ballot(ent != SLOT_INVALID && t.state == TRAV_DONE));
                if (n_ready || n_shade || !exhausted || finished * RUN_TAIL_DIV >= (uint32_t)__popcll(busy)) break;
            }
        }
#ifdef GD_RUN_TIMELINE
        { const unsigned long long now = wall_clock64(); if (exhausted) tl_tail_trav += now - tl_mark; tl_mark = now; }
#endif
    }
    if (COUNT) flush_counters(wc, segments, gcounters);
#ifdef GD_RUN_TIMELINE
    if (lane_id() == 0 && blockIdx.x < 8192) {
        unsigned long long *o = g_run_timeline + 24 * blockIdx.x;
        o[12] = tl_m2; o[13] = tl_m4; o[14] = tl_r2; o[15] = tl_r4;
        o[16] = tp_box; o[17] = tp_nbox; o[18] = tp_leaf; o[19] = tp_nleaf; o[20] = tp_loop; o[21] = tp_busy; o[22] = 0; o[23] = 0;
        o[6] = tl_nready; o[7] = tl_nshade; o[8] = tl_tail_trav; o[9] = tl_tail_other; o[10] = tl_tail_rounds; o[11] = tl_tail_trig;
        o[0] = tl_start; o[1] = tl_dry; o[2] = wall_clock64(); o[3] = tl_lanes; o[4] = tl_rounds;
        for (int k = 0; k < 256; k++) if (tl_hist[k]) atomicAdd(&g_run_hist[k], (unsigned long long)tl_hist[k]);
        o[5] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);  // XCC_ID, HW_ID
    }
#endif
}

}  // namespace
