// kernels_pipeline.h — the render kernels of libgpuart_hip.so (gfx950): the wavefront path-tracing pipeline
// (k_gen, k_trace, k_shade, k_accumulate) and the one-thread-per-pixel kernels (k_direct, k_pt_mega).
// Included by gpuart_hip.hip only; DESIGN.md section 4 describes the pipeline.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_shade.h"
#include "gpuart_hip.h"

using namespace gd;

#define BLOCK 64          // one wavefront per workgroup
#define SHADE_ROUNDS 8    // queue entries per lane a shading wave handles between two queue appends

/// Scheduling knobs of the persistent BVH-query kernel (defaults chosen by tools/sweep.py on MI355X;
/// overridable through GPUART_HIP_* environment variables for tuning runs; results never depend on them).
struct TraceTuning {
    uint32_t chunk;         ///< rays a wave takes from a queue per fetch
    uint32_t refill_lanes;  ///< a wave goes back for new rays once this many lanes are idle
    uint32_t leaf_lanes;    ///< primitive tests are issued once this many lanes wait at a leaf ...
    uint32_t leaf_share;    ///< ... or 1/leaf_share of the lanes that still hold a ray
    uint32_t xcd_queues;    ///< experiment (GPUART_HIP_XCD_QUEUES, default 0): the waves of XCD x (workgroup index mod 8) serve the x-th
                            ///< eighth of the ray queue — contiguous slots, one region of the image — from a cursor of their own, so
                            ///< that an XCD's L2 sees one eighth of the rays' working set; no stealing between the eighths
};

// =================================================================================================
// Kernels
// =================================================================================================
namespace {

#ifdef GD_STEP_STATS
__device__ unsigned long long g_step_stats[8];
__device__ unsigned long long g_phase_ticks[4];  // k_trace: summed over waves, shader-clock ticks in refill / traverse / settle + retire
#endif

/// Per-path state of one run of the wavefront pipeline: `batch` passes x `n_slots` pixel slots (pixel slot p: 8x8 tile
/// p/64 of the tile in row-major tile order, pixel p%64 inside it), all arrays SoA and 16-byte aligned.
/// Path slot s = pixel slot x batch + pass (slot_pass / slot_pixel_slot below): the passes of a run are INTERLEAVED per
/// pixel, so that the 64 lanes of a wave hold few pixels x several passes. The camera ray of a pixel differs between
/// passes only by its sub-pixel jitter, and so does the Sun-shadow ray of its first hit: such lanes visit the same nodes,
/// the vector L1 serves identical addresses of one instruction with one access, and L1 accesses are what bounds the BVH
/// queries. Path arithmetic does not depend on the slot order: results are unchanged bit for bit.
struct PathBuffers {
    float4 *ray_o, *ray_d;   ///< ray of the current / next segment
    uint2 *hit;              ///< closest-hit result of the current segment: (bits(t), primitive index)
    float4 *cw;              ///< colorWeight
    float4 *pc;              ///< pathColor
    float4 *sun;             ///< pending Sun term (xyz), w = bits(1: the path ends after its shadow query)
    float4 *color;           ///< colour of the earlier paths of this pass (NumPathsPerPixel > 1)
    uint32_t *queue[2];      ///< slots that trace segment s (ping-pong)
    uint32_t *shadow_queue;  ///< slots with a pending Sun shadow query
    uint32_t *counters;      ///< per segment s: [4s] closest-hit rays, [4s+1] k_trace's fetch cursor (for the launch whose
                             ///< closest part is s), [4s+2] shadow rays, [4s+3] fetch cursor of a shadow-only launch
    uint32_t *xcd_cursors;   ///< [16s + x] / [16s + 8 + x]: the same two cursors per XCD x (TraceTuning::xcd_queues)
    uint32_t n_slots;        ///< path slots per pass (pixels of the tile, 8x8-tile padded)
    uint32_t batch;          ///< passes processed together: slot s belongs to pass s / n_slots, pixel slot s % n_slots
    uint32_t tile_pixels;    ///< stride between the passes' colour planes in `passcolor`
};

#define MAX_BATCH 64
/// RandSeed of every pass of a pipeline run (slot = pass x pixel; up to 16M paths or MAX_BATCH passes per run).
struct SeedBatch {
    float4 seed[MAX_BATCH];
};
#define SLOT_INVALID 0xffffffffu

GD_FN int lane_id() { return threadIdx.x & 63; }
/// Set bits of a ballot below this lane (v_mbcnt: no 64-bit lane mask to keep in registers).
GD_FN uint32_t rank_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

/// Pass (index into the run's RandSeed batch / colour planes) and pixel slot of a path slot.
GD_FN uint32_t slot_pass(const PathBuffers &b, uint32_t slot) { return slot % b.batch; }
GD_FN uint32_t slot_pixel_slot(const PathBuffers &b, uint32_t slot) { return slot / b.batch; }

GD_FN void flush_counters(const WorkCounters &wc, uint32_t segments, unsigned long long *g) {
    // one atomic per counter per wavefront
    uint32_t v[10] = {wc.rays, wc.nodes, wc.prims[0], wc.prims[1], wc.prims[2], wc.prims[3], segments, wc.steps, wc.steps_top, wc.rewalks};
    for (int k = 0; k < 10; k++) {
        unsigned long long s = v[k];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane_id() == 0 && s) atomicAdd(&g[k], s);
    }
}

/// slot -> pixel of the tile; false for the padding slots of ragged edge tiles.
GD_FN bool slot_pixel(const Frame &f, uint32_t slot, uint32_t &lx, uint32_t &ly) {
    uint32_t tiles_x = (f.tw + 7) / 8;
    uint32_t t = slot >> 6, w = slot & 63;
    if (f.tile_order) t = f.tile_order[t];
    lx = (t % tiles_x) * 8 + (w & 7);
    ly = (t / tiles_x) * 8 + (w >> 3);
    return lx < f.tw && ly < f.th;
}

/// A run that gathers the cost of the tile's 8x8 blocks (Frame::tile_cost, wave-uniform): one count per shaded path segment.
GD_FN void tile_cost_add(const Frame &f, const PathBuffers &b, uint32_t slot) {
    if (f.tile_cost) {
        uint32_t t = slot_pixel_slot(b, slot) >> 6;
        if (f.tile_order) t = f.tile_order[t];
        atomicAdd(&f.tile_cost[t], 1u);
    }
}

/// The birth order of the paths of the runs to come: the tile's 8x8 pixel blocks sorted by the cost the last gathering run counted
/// for them, most expensive class first, row-major within a class (neighbouring blocks stay neighbours: they visit the same part of
/// the tree). Why: a run ends when its LAST path ends, and a path is a chain of up to five dependent closest-hit queries — born
/// last, a long path runs on with the machine nearly empty; born first, it overlaps with everything else, and the run drains on the
/// one-segment paths of the sky (longest-processing-time-first). No pixel's value depends on the order. One workgroup; counts are
/// quantised to TO_CLASSES classes of the largest count, sorted by counting (stable), and zeroed for the next gathering run.
#define TO_THREADS 512
#define TO_CLASSES 24
__global__ void __launch_bounds__(TO_THREADS) k_tile_order(uint32_t *__restrict__ cost, uint32_t *__restrict__ order, uint32_t tiles) {
    __shared__ uint32_t cnt[TO_CLASSES][TO_THREADS];
    __shared__ uint32_t red[TO_THREADS];
    __shared__ uint32_t class_base[TO_CLASSES];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (tiles + TO_THREADS - 1) / TO_THREADS;  // thread `tid` owns the blocks [lo, hi): in order, so that the sort is stable
    const uint32_t lo = min(tid * per, tiles), hi = min(lo + per, tiles);
    uint32_t mx = 0;
    for (uint32_t i = lo; i < hi; i++) mx = max(mx, cost[i]);
    red[tid] = mx;
    __syncthreads();
    for (uint32_t s = TO_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = max(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    auto cls = [mx](uint32_t v) { return (uint32_t)(TO_CLASSES - 1) - (uint32_t)(((unsigned long long)v * TO_CLASSES) / ((unsigned long long)mx + 1)); };  // 0: the most expensive
    for (uint32_t c = 0; c < TO_CLASSES; c++) cnt[c][tid] = 0;
    for (uint32_t i = lo; i < hi; i++) cnt[cls(cost[i])][tid]++;
    __syncthreads();
    if (tid < TO_CLASSES) {  // within a class: the blocks of thread 0, then thread 1, ...
        uint32_t run = 0;
        for (uint32_t k = 0; k < TO_THREADS; k++) { const uint32_t v = cnt[tid][k]; cnt[tid][k] = run; run += v; }
        class_base[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t c = 0; c < TO_CLASSES; c++) { const uint32_t v = class_base[c]; class_base[c] = run; run += v; }
    }
    __syncthreads();
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t c = cls(cost[i]);
        order[class_base[c] + cnt[c][tid]++] = i;
        cost[i] = 0;
    }
}

/// Lane 0's value in every lane, where all 64 lanes are active (the persistent loops' wave-uniform control flow). GD_UNIFORM = 1: through
/// v_readfirstlane — a SCALAR the compiler knows to be the same in every lane, so that what is derived from it (the chunk bounds, `exhausted`)
/// lives in scalar registers and wave-uniform branches on it are scalar branches; a shuffle (ds_bpermute, the code of rounds 1-5: 0) yields
/// a vector value, the loop exits that test it became execution-mask arithmetic on both paths.
#ifndef GD_UNIFORM
#define GD_UNIFORM 1
#endif
GD_FN uint32_t wave_value(uint32_t v) {
#if GD_UNIFORM
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return (uint32_t)__shfl(v, 0, 64);
#endif
}

/// The traversal stack of this lane: its column of the wave's LDS ring, its column of the launch's spill area
/// (`wave` of `total_lanes / 64` waves).
GD_FN TravStack make_stack(uint2 *ring_a, float *ring_b, uint4 *spill, uint32_t total_lanes, uint32_t wave, uint32_t column) {
    TravStack st;
    st.ring_a = ring_a + column;
    st.ring_b = ring_b + column;
    st.ring_stride = BLOCK;
    st.spill = spill + (size_t)wave * BLOCK + column;
    st.spill_stride = total_lanes;
    st.reset();
    return st;
}
GD_FN TravStack make_stack(uint2 *ring_a, float *ring_b, uint4 *spill, uint32_t total_lanes, uint32_t wave) {
    return make_stack(ring_a, ring_b, spill, total_lanes, wave, (uint32_t)lane_id());
}
GD_FN TravStack make_stack(uint2 *ring_a, float *ring_b, uint4 *spill, uint32_t total_lanes) {
    return make_stack(ring_a, ring_b, spill, total_lanes, blockIdx.x);
}

/// Thin-wave modes (device_scene.h): regroups a wave's rays for `to` lanes per ray. The q-th ray in flight (`flying`: the first
/// lanes of the groups that hold one) moves to the lanes [to q, to q + to) as identical replicas — the two tags, the ray, the
/// traversal state — and so does its stack column (LDS ring and global spill): a group's column is that of its first lane in
/// every mode. Groups beyond the rays in flight come out idle (state DONE, empty stack; their tags are the caller's to reset).
/// `xfer`: 64 words of LDS scratch.
GD_FN void thin_regroup(uint32_t to, unsigned long long flying, uint32_t *xfer, uint2 *ring_a, float *ring_b, uint4 *spill,
                        uint32_t &tag0, uint32_t &tag1, F3 &ro, F3 &rd, F3 &rdiv, Trav &t, TravStack &st, int *src_lane = nullptr) {
    if ((flying >> lane_id()) & 1ull) xfer[__popcll(flying & ((1ull << lane_id()) - 1))] = (uint32_t)lane_id();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const uint32_t q = (uint32_t)lane_id() / to;
    const bool holds = q < (uint32_t)__popcll(flying);
    const int src = holds ? (int)xfer[q] : lane_id();
    if (src_lane) *src_lane = src;  // for whatever else the caller keeps per ray
    tag0 = __shfl(tag0, src, 64); tag1 = __shfl(tag1, src, 64);
    ro = f3(__shfl(ro.x, src, 64), __shfl(ro.y, src, 64), __shfl(ro.z, src, 64));
    rd = f3(__shfl(rd.x, src, 64), __shfl(rd.y, src, 64), __shfl(rd.z, src, 64));
    rdiv = f3(__shfl(rdiv.x, src, 64), __shfl(rdiv.y, src, 64), __shfl(rdiv.z, src, 64));
    t.closest = __shfl(t.closest, src, 64); t.hit_prim = __shfl(t.hit_prim, src, 64); t.node = __shfl(t.node, src, 64);
    t.entry = __shfl(t.entry, src, 64); t.state = __shfl(t.state, src, 64); t.second = __shfl(t.second, src, 64);
    uint32_t sp = __shfl(st.sp, src, 64), base = __shfl(st.base, src, 64);
    if (!holds) { t.state = TRAV_DONE; sp = 0; base = 0; }
    // Columns, row by row: the loads of a row are one instruction of the whole wave and precede its stores, and a column that
    // is some ray's destination may be another ray's source — but never in a different row.
    const bool mover = holds && ((uint32_t)lane_id() % to) == 0 && src != lane_id();
    for (uint32_t k = 0; k < GD_RING; k++) {
        const uint2 a = ring_a[k * BLOCK + (uint32_t)src];
        const float bb = ring_b[k * BLOCK + (uint32_t)src];
        if (mover) { ring_a[k * BLOCK + lane_id()] = a; ring_b[k * BLOCK + lane_id()] = bb; }
    }
    for (uint32_t k = 0; __ballot(mover && k < base) != 0; k++) {  // the spilled part (deep trees only)
        const bool mv = mover && k < base;
        uint4 v = make_uint4(0, 0, 0, 0);
        const size_t row = (size_t)k * gridDim.x * BLOCK + (size_t)blockIdx.x * BLOCK;
        if (mv) v = spill[row + (uint32_t)src];
        if (mv) spill[row + lane_id()] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK, blockIdx.x, (uint32_t)lane_id() / to * to);
    st.sp = sp; st.base = base;
}

/// Wave-aggregated append: lanes with `pred` get consecutive positions of `queue` (one atomic per wave).
GD_FN void queue_push(uint32_t *queue, uint32_t *counter, bool pred, uint32_t value) {
    unsigned long long m = __ballot(pred);
    if (!m) return;
    uint32_t base = 0;
    int leader = __ffsll((long long)m) - 1;
    if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (pred) queue[base + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1))] = value;
}

/// Adds a finished path's colour to its pixel's colour of this pass (path_tracing.glsl:252: color += pathColor for
/// every path of the pass). The last path stores the pass colour in `passcolor` (tile row-major); k_accumulate then
/// performs path_tracing.glsl:255, accum = PrevRadiance + color, in pass order.
GD_FN void path_commit(const Frame &f, const PathBuffers &b, float4 *passcolor, uint32_t slot, int j, int npaths, F3 value) {
    F3 c = (j == 0) ? f3(0.0f + value.x, 0.0f + value.y, 0.0f + value.z) : xyz(b.color[slot]) + value;
    if (j == npaths - 1) {
        uint32_t lx, ly;
        slot_pixel(f, slot_pixel_slot(b, slot), lx, ly);
        passcolor[(size_t)slot_pass(b, slot) * b.tile_pixels + (size_t)ly * f.tw + lx] = make_float4(c.x, c.y, c.z, 0);
    } else
        b.color[slot] = make_float4(c.x, c.y, c.z, 0);
}

/// path_tracing.glsl:255 for one finished pass: accum = PrevRadiance + color. Launched in pass order on the
/// context's primary stream, so float additions happen in the reference's order whatever the overlap of passes.
__global__ void k_accumulate(float4 *__restrict__ accum, const float4 *__restrict__ passcolor, size_t n, uint32_t batch) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float4 a = accum[i];
        for (uint32_t k = 0; k < batch; k++) {  // the passes of a batch, oldest first
            float4 c = passcolor[(size_t)k * n + i];
            a = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w);
        }
        accum[i] = a;
    }
}

// ---- wavefront stage 0: first ray of path j of every pixel (path_tracing.glsl:141-175) ---------------
__global__ void __launch_bounds__(BLOCK) k_gen(Frame f, gpuart_params P, SeedBatch seeds, int j, int npaths, PathBuffers b,
                                               float4 *accum) {
    const bool no_segments = !(P.maxSegments > 0 && 1.0f > P.minWeight);
    const uint32_t total = b.n_slots * b.batch;
    for (uint32_t slot = blockIdx.x * BLOCK + threadIdx.x; slot < total; slot += gridDim.x * BLOCK) {
        uint32_t lx, ly;
        bool valid = slot_pixel(f, slot_pixel_slot(b, slot), lx, ly);
        const float4 seed = seeds.seed[slot_pass(b, slot)];
        uint32_t q = SLOT_INVALID;
        if (valid) {
            F3 rs0, rd0, rs, rd;
            camera_ray(f, f.x0 + lx, frame_y(f, ly), rs0, rd0);
            if (no_segments) {  // the GLSL loop body never runs: i == 0 and no user-sphere hit
                path_commit(f, b, accum, slot, j, npaths, path_finish(P, rd0, 0, false, false, f3(0, 0, 0)));
            } else {
                path_begin(P, seed, j, rs0, rd0, rs, rd);
                b.ray_o[slot] = make_float4(rs.x, rs.y, rs.z, 0);
                b.ray_d[slot] = make_float4(rd.x, rd.y, rd.z, 0);
                q = slot;  // (colorWeight = 1 and pathColor = 0 are not stored: k_shade knows them for segment 0)
            }
        }
        b.queue[0][slot] = q;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) b.counters[0] = no_segments ? 0u : total;
}

// ---- wavefront stage 1/3: BVH queries, persistent waves with lane refill ----------------------------------
// One launch serves two queues at once: the closest-hit queries of segment `seg_c` (slots of queue[seg_c&1]; result ->
// hit[slot]) and the Sun-shadow queries of segment `seg_s` (slots of shadow_queue; they apply the Sun term and, for
// paths that end with that segment, commit the path) — after shading segment s both the shadow queries of s and the
// closest-hit queries of s+1 are ready and touch disjoint data. Either may be absent (-1). The long closest-hit rays
// come first in the combined index space, the short shadow rays (they stop at the first accepted hit when
// `any_shadow`) fill the end of the launch.
// A lane that finishes its ray takes the next one from the queue (wave-local chunk, one atomic per chunk), so all
// 64 lanes keep traversing; box tests and leaf tests are issued as separate wave-wide phases (leaf code waits until
// LEAF_LANES lanes need it).
#define GD_FLAT_TYPES ((1 << gd::P_DISC) | (1 << gd::P_TRIANGLE))
#define GD_ROUND_TYPES ((1 << gd::P_SPHERE) | (1 << gd::P_DISC))  // sphere scenes (Scene P, the cluster): no triangle, no cone code
#define GD_LEAN_TYPES(T) (((T) & ~GD_REF_ORDER) == GD_FLAT_TYPES || ((T) & ~GD_REF_ORDER) == GD_ROUND_TYPES)
#ifndef GD_TRACE_WAVES
#define GD_TRACE_WAVES 5  // waves per SIMD the register allocation must allow (<= 96 VGPRs)
#endif
#ifndef GD_TRACE_THIN
#define GD_TRACE_THIN 4  // most lanes per ray in a draining wave (1: never leave wide mode)
#endif
#ifndef GD_TRACE_WAVES_LEAN
#define GD_TRACE_WAVES_LEAN 6  // the kernels without cone / sphere code fit 6 waves per SIMD (<= 80 VGPRs)
#endif
template <bool COUNT, int TYPES>
__global__ void __launch_bounds__(BLOCK, GD_LEAN_TYPES(TYPES) ? GD_TRACE_WAVES_LEAN : GD_TRACE_WAVES)
k_trace(Scene sc, Frame f, gpuart_params P, PathBuffers b, int seg_c, int seg_s, int any_shadow, int j, int npaths, float4 *accum,
        uint4 *spill, unsigned long long *gcounters, TraceTuning tune) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    const uint32_t wave_id = blockIdx.x, n_waves = gridDim.x;  // one wavefront per workgroup
    TravStack st = make_stack(ring_a, ring_b, spill, n_waves * BLOCK, wave_id);
    // Thin-wave modes (device_scene.h): once the queue is empty and the wave is down to 32 (16) rays, pairs (quads) of lanes carry
    // them. Not in the counting variant (its counters are per lane) nor for trees with irregular boxes.
    constexpr bool THIN_OK = GD_TRACE_THIN > 1 && !COUNT && GD_BOXES_OF(TYPES) == GD_BOXES_FAST;
    // closest-hit queries enter the nearer child first (device_scene.h, GD_NEAREST): fewer node visits; a query whose answer that
    // walk cannot certify goes round again in the reference's order (`refwalk`)
    constexpr bool NEAR = GD_NEAREST_OF(TYPES) && !COUNT;
    uint32_t M = 1, sub = 0;  // M wave-uniform
    const uint32_t *queue_c = b.queue[seg_c & 1];
    const uint32_t n_c = seg_c >= 0 ? b.counters[4 * seg_c] : 0u;
    const uint32_t n = n_c + (seg_s >= 0 ? b.counters[4 * seg_s + 2] : 0u);
    uint32_t *cursor = &b.counters[seg_c >= 0 ? 4 * seg_c + 1 : 4 * seg_s + 3];
    const F3 sun = f3(P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]);
    WorkCounters wc = {0, 0, {0, 0, 0, 0}, 0, 0, 0};

    // The first chunk of every wave is static (chunk index = workgroup index); later chunks come from the
    // shared cursor, which therefore starts behind the static ones. No atomic at all for small queues.
    uint32_t static_end = n_waves * tune.chunk;
    uint32_t chunk_next = min(wave_id * tune.chunk, n), chunk_end = min((wave_id + 1) * tune.chunk, n);  // wave-uniform
    uint32_t q_end = n;  // end of the part of the queue this wave serves
    if (tune.xcd_queues) {  // (experiment) this wave's XCD serves its own eighth of the queue
        const uint32_t xcd = wave_id & 7u, wx = wave_id >> 3, nwx = (n_waves + 7u - xcd) >> 3;
        const uint32_t lo = (uint32_t)((unsigned long long)n * xcd / 8u);
        q_end = (uint32_t)((unsigned long long)n * (xcd + 1u) / 8u);
        chunk_next = min(lo + wx * tune.chunk, q_end); chunk_end = min(lo + (wx + 1) * tune.chunk, q_end);
        static_end = lo + nwx * tune.chunk;
        cursor = &b.xcd_cursors[16 * (seg_c >= 0 ? seg_c : seg_s) + (seg_c >= 0 ? 0 : 8) + xcd];
    }
    bool exhausted = false;                                                                                   // wave-uniform
    uint32_t slot = SLOT_INVALID;
    bool shadow = false;                    // this lane's ray is a Sun-shadow query
    bool refwalk = false;                   // this lane's query is on its second walk, in the reference's order
    F3 ro = f3(0, 0, 0), rd = f3(1, 0, 0);  // the ray (two F3 locals: a long-lived Ray aggregate ends up in scratch)
    F3 rdiv = f3(1, 1, 1);
    Trav t; t.state = TRAV_DONE; t.closest = 0; t.hit_prim = GD_NO_PRIM; t.node = 0; t.entry = 0; t.second = 0;

#ifdef GD_STEP_STATS
    // diagnostic build (never the product): how full the wave's steps are — box steps and the lanes in them, leaf steps and the lanes in
    // them, rounds of the wide loop and the lanes that hold a ray in them, refill episodes (tools/step_stats.py)
    uint32_t ss_box = 0, ss_box_lanes = 0, ss_leaf = 0, ss_leaf_lanes = 0, ss_rounds = 0, ss_held = 0, ss_refills = 0;
    unsigned long long ss_t_refill = 0, ss_t_trav = 0, ss_t_retire = 0, ss_mark = __builtin_amdgcn_s_memtime();  // shader-clock ticks per phase of the outer loop
#define GD_SS_PHASE(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - ss_mark; ss_mark = now_; }
#else
#define GD_SS_PHASE(acc)
#endif
    for (;;) {
        // ---- refill idle lanes from the queue
        unsigned long long idle = __ballot(slot == SLOT_INVALID);
#ifdef GD_STEP_STATS
        ss_refills++;
#endif
        while (idle && !exhausted) {
            if (chunk_next == chunk_end) {
                if (static_end >= q_end) { exhausted = true; break; }
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(cursor, tune.chunk);
                base = wave_value(base) + static_end;
                if (base >= q_end) { exhausted = true; break; }
                chunk_next = base;
                chunk_end = min(base + tune.chunk, q_end);
            }
            uint32_t want = (uint32_t)__popcll(idle), take = min(want, chunk_end - chunk_next);
            uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane_id()) - 1));
            if (slot == SLOT_INVALID && rank < take) {
                const uint32_t i = chunk_next + rank;
                const bool sh = i >= n_c;
                uint32_t s = sh ? b.shadow_queue[i - n_c] : queue_c[i];
                if (s != SLOT_INVALID) {
                    slot = s;
                    shadow = sh;
                    refwalk = false;
                    ro = xyz(b.ray_o[s]);
                    rd = sh ? sun : xyz(b.ray_d[s]);
                    rdiv = f3(1 / rd.x, 1 / rd.y, 1 / rd.z);
                    trav_init<GD_BOXES_OF(TYPES)>(sc, Ray{ro, rd}, rdiv, t, st, &wc, COUNT, NEAR && !sh);
                }
            }
            chunk_next += take;
            idle = __ballot(slot == SLOT_INVALID);
            if (take == want) break;
        }
        const unsigned long long flying = __ballot(slot != SLOT_INVALID && sub == 0);
        GD_SS_PHASE(ss_t_refill)
        if (flying == 0) {
            if (exhausted) break;
            continue;
        }
        if (THIN_OK && exhausted && M < (uint32_t)GD_TRACE_THIN) {
            const uint32_t left = (uint32_t)__popcll(flying);
            const uint32_t to = left <= BLOCK / 4 && GD_TRACE_THIN >= 4 ? 4u : left <= BLOCK / 2 ? 2u : 1u;
            if (to > M) {
                uint32_t sh = (shadow ? 1u : 0u) | (refwalk ? 2u : 0u);
                // (the stack ring doubles as the scratch of the move: row 0 is copied first, and the table is read before that)
                __shared__ uint32_t xfer[BLOCK];
                thin_regroup(to, flying, xfer, ring_a, ring_b, spill, slot, sh, ro, rd, rdiv, t, st);
                shadow = (sh & 1u) != 0; refwalk = (sh & 2u) != 0;
                if ((uint32_t)lane_id() / to >= left) slot = SLOT_INVALID;
                M = to;
                sub = (uint32_t)lane_id() & (M - 1);
            }
        }
        if (THIN_OK && M > 1) {
            // the loop below with M lanes per ray (every ray is in flight until it is done: the queue is empty)
            auto thin_rounds = [&](auto width) {
                constexpr int W = decltype(width)::value;
                constexpr unsigned long long LEAD = W == 4 ? 0x1111111111111111ull : 0x5555555555555555ull;
                for (;;) {
                    // (no L1-warming loads for stacked children here, unlike k_run: the end of a k_trace launch overlaps the launches of
                    //  other pass lanes, the memory system is busy, and the extra requests cost 6 % of a pass — profiles/r04/thin_prefetch.txt)
                    if (t.state == TRAV_DESCEND) trav_step_box_thin<W, NEAR>(sc, ro, rd, rdiv, t, st, sub, !shadow && !refwalk);
                    const unsigned long long at_leaf = __ballot((t.state & 1) != 0) & LEAD;
                    unsigned long long busy = __ballot(t.state != TRAV_DONE) & LEAD;
                    const uint32_t waiting = (uint32_t)__popcll(at_leaf);
                    if (at_leaf && (W * waiting >= tune.leaf_lanes || tune.leaf_share * waiting >= (uint32_t)__popcll(busy))) {
                        if (t.state & 1) {
                            trav_step_leaf_thin<W, TYPES, NEAR>(sc, ro, rd, t, st, sub, !shadow && !refwalk);
                            if (shadow && any_shadow && t.hit_prim != GD_NO_PRIM) t.state = TRAV_DONE;
                        }
                        busy = __ballot(t.state != TRAV_DONE) & LEAD;
                    }
                    if (!busy) break;
                    // a wave may move to quads once it is down to 16 rays
                    if (W == 2 && GD_TRACE_THIN >= 4 && (uint32_t)__popcll(busy) <= BLOCK / 4) break;
                }
            };
            if (M == 2) thin_rounds(std::integral_constant<int, 2>());
            else thin_rounds(std::integral_constant<int, 4>());
        } else
        // ---- traverse until enough lanes have finished (a lane without a ray is in state DONE)
        for (;;) {
#ifdef GD_STEP_STATS
            { const unsigned long long dd = __ballot(t.state == TRAV_DESCEND); if (dd) { ss_box++; ss_box_lanes += (uint32_t)__popcll(dd); } ss_rounds++; ss_held += (uint32_t)__popcll(__ballot(slot != SLOT_INVALID)); }
#endif
            if (t.state == TRAV_DESCEND) trav_step_box<COUNT, GD_BOXES_OF(TYPES), NEAR>(sc, Ray{ro, rd}, rdiv, t, st, COUNT ? &wc : nullptr, !shadow && !refwalk);
            unsigned long long at_leaf = __ballot((t.state & 1) != 0);  // the leaf states are the odd ones
            unsigned long long descending = __ballot(t.state == TRAV_DESCEND);
            // leaves are tested once 1/leaf_share of the lanes that still have a ray wait at one (at most leaf_lanes):
            // a wave that is draining its last rays must not hold leaves back for a quorum it can no longer reach
            const uint32_t waiting = (uint32_t)__popcll(at_leaf);
            if (at_leaf && (waiting >= tune.leaf_lanes || tune.leaf_share * waiting >= waiting + (uint32_t)__popcll(descending))) {
#ifdef GD_STEP_STATS
                ss_leaf++; ss_leaf_lanes += waiting;
#endif
                if (t.state & 1) {
                    trav_step_leaf<false, COUNT, TYPES, NEAR>(sc, Ray{ro, rd}, t, st, COUNT ? &wc : nullptr, !shadow && !refwalk);
                    // the reference only asks a shadow query whether anything was hit: one accepted hit settles it
                    if (shadow && any_shadow && t.hit_prim != GD_NO_PRIM) t.state = TRAV_DONE;
                }
                descending = __ballot(t.state == TRAV_DESCEND);
                at_leaf = __ballot((t.state & 1) != 0);
            }
            unsigned long long busy = descending | at_leaf;
            if (!busy) break;
            if (!exhausted && 64u - (uint32_t)__popcll(busy) >= tune.refill_lanes) break;
            // a draining wave may move to pairs / quads once it is down to 32 rays
            if (THIN_OK && exhausted && (uint32_t)__popcll(busy) <= BLOCK / 2) break;
        }
        GD_SS_PHASE(ss_t_trav)
        // ---- a finished nearest-first query that cannot vouch for its answer walks again, in the reference's order (every replica alike)
        if (NEAR && slot != SLOT_INVALID && t.state == TRAV_DONE && trav_settle<NEAR>(t, !shadow && !refwalk)) {
            refwalk = true;
            trav_init<GD_BOXES_OF(TYPES)>(sc, Ray{ro, rd}, rdiv, t, st, &wc, false);
        }
        // ---- retire finished rays (a ray's first replica retires it)
        if (slot != SLOT_INVALID && t.state == TRAV_DONE && (!THIN_OK || sub == 0)) {
            if (!shadow) {
                b.hit[slot] = make_uint2(__float_as_uint(t.closest), t.hit_prim);
            } else {
                float4 term = b.sun[slot];
                F3 pathColor = xyz(b.pc[slot]);
                if (sun_visible(P, ro, sun, t.hit_prim)) pathColor = pathColor + xyz(term);
                if (__float_as_uint(term.w) & 1u) path_commit(f, b, accum, slot, j, npaths, pathColor);
                else b.pc[slot] = make_float4(pathColor.x, pathColor.y, pathColor.z, 0);
            }
        }
        if (slot != SLOT_INVALID && t.state == TRAV_DONE) slot = SLOT_INVALID;
        GD_SS_PHASE(ss_t_retire)
    }
    if (COUNT) flush_counters(wc, 0, gcounters);
#ifdef GD_STEP_STATS
    if (lane_id() == 0) {
        const uint32_t v[7] = {ss_box, ss_box_lanes, ss_leaf, ss_leaf_lanes, ss_rounds, ss_held, ss_refills};
        for (int k = 0; k < 7; k++) atomicAdd(&g_step_stats[k], (unsigned long long)v[k]);
        atomicAdd(&g_phase_ticks[0], ss_t_refill); atomicAdd(&g_phase_ticks[1], ss_t_trav); atomicAdd(&g_phase_ticks[2], ss_t_retire);
    }
#endif
}

// ---- wavefront stage 2: shade segment `seg` of every path in queue[seg&1] (path_tracing.glsl:182-233) ---
template <bool REFWORK>
__global__ void __launch_bounds__(BLOCK) k_shade(Scene sc, Frame f, gpuart_params P, SeedBatch seeds, PathBuffers b, int seg,
                                                 int nseg, int j, int npaths, float4 *accum, unsigned long long *gcounters) {
    // Survivors are appended to the next queues through a per-wave staging list in LDS that is flushed with
    // ONE atomic per SHADE_ROUNDS*64 processed paths: a single atomic word sustains only ~90 appends/us on
    // MI355X, and one append per wave per 64 paths made this kernel atomic-bound.
    __shared__ uint32_t stage_next[SHADE_ROUNDS * BLOCK], stage_shadow[SHADE_ROUNDS * BLOCK];
    const uint32_t n = b.counters[4 * seg];
    const uint32_t *queue = b.queue[seg & 1];
    uint32_t *next_queue = b.queue[(seg + 1) & 1];
    uint32_t segments = 0;
    const uint32_t span = SHADE_ROUNDS * BLOCK;                 // consecutive queue entries one wave handles at a time
    const uint32_t spans = (n + span - 1) / span;
    for (uint32_t sp = blockIdx.x; sp < spans; sp += gridDim.x) {
      uint32_t n_next = 0, n_shadow = 0;                        // wave-uniform fill of the staging lists
      for (uint32_t k = 0; k < SHADE_ROUNDS; k++) {
        uint32_t e = sp * span + k * BLOCK + threadIdx.x;
        uint32_t slot = e < n ? queue[e] : SLOT_INVALID;
        bool go_on = false, shadow = false;
        if (slot != SLOT_INVALID) {
            tile_cost_add(f, b, slot);
            Ray r; r.o = xyz(b.ray_o[slot]); r.d = xyz(b.ray_d[slot]);
            uint2 h = b.hit[slot];
            F3 cw = f3(1, 1, 1), pathColor = f3(0, 0, 0);  // path_tracing.glsl:156-157
            if (seg > 0) { cw = xyz(b.cw[slot]); pathColor = xyz(b.pc[slot]); }
            F3 rstart = r.o, rdir = r.d;
            segments++;
            const float4 seed = seeds.seed[slot_pass(b, slot)];
            ShadeResult s = path_shade(sc, P, seed, seg, r, __uint_as_float(h.x), h.y, rstart, rdir, cw, pathColor);
            if (s.broke) {
                uint32_t lx, ly; F3 rs0, rd0;
                slot_pixel(f, slot_pixel_slot(b, slot), lx, ly);
                camera_ray(f, f.x0 + lx, frame_y(f, ly), rs0, rd0);
                path_commit(f, b, accum, slot, j, npaths, path_finish(P, rd0, seg, s.ush, s.specular, pathColor));
            } else {
                // `seg + 1 < nseg` can only fail if the host's segment bound were too small: such a path is committed
                // here instead of being queued for a launch that will not come (it would keep a stale colour)
                go_on = s.next == PATH_CONTINUES && seg + 1 < nseg;
                shadow = s.want_shadow && (REFWORK || s.sun_matters);
                if (shadow)
                    b.sun[slot] = make_float4(s.sun_term.x, s.sun_term.y, s.sun_term.z, __uint_as_float(go_on ? 0u : 1u));
                if (go_on || shadow) b.ray_o[slot] = make_float4(rstart.x, rstart.y, rstart.z, 0);
                if (go_on) {
                    b.ray_d[slot] = make_float4(rdir.x, rdir.y, rdir.z, 0);
                    b.cw[slot] = make_float4(cw.x, cw.y, cw.z, 0);
                }
                if (go_on || shadow) b.pc[slot] = make_float4(pathColor.x, pathColor.y, pathColor.z, 0);
                if (!go_on && !shadow) path_commit(f, b, accum, slot, j, npaths, pathColor);  // i >= 1: no special case
            }
        }
        unsigned long long m1 = __ballot(go_on), m2 = __ballot(shadow);
        unsigned long long below = (1ull << lane_id()) - 1;
        if (go_on) stage_next[n_next + (uint32_t)__popcll(m1 & below)] = slot;
        if (shadow) stage_shadow[n_shadow + (uint32_t)__popcll(m2 & below)] = slot;
        n_next += (uint32_t)__popcll(m1);
        n_shadow += (uint32_t)__popcll(m2);
      }
      // flush: one atomic per list, then a coalesced copy
      __syncthreads();
      uint32_t base1 = 0, base2 = 0;
      if (lane_id() == 0) {
          if (n_next) base1 = atomicAdd(&b.counters[4 * (seg + 1)], n_next);
          if (n_shadow) base2 = atomicAdd(&b.counters[4 * seg + 2], n_shadow);
      }
      base1 = wave_value(base1);
      base2 = wave_value(base2);
      for (uint32_t i = threadIdx.x; i < n_next; i += BLOCK) next_queue[base1 + i] = stage_next[i];
      for (uint32_t i = threadIdx.x; i < n_shadow; i += BLOCK) b.shadow_queue[base2 + i] = stage_shadow[i];
      __syncthreads();
    }
    if (REFWORK) {
        WorkCounters z = {0, 0, {0, 0, 0, 0}, 0, 0, 0};
        flush_counters(z, segments, gcounters);
    }
}

// ---- megakernels: one thread per pixel, persistent grid over 8x8 tiles -------------------------------------
template <bool REFWORK>
__global__ void __launch_bounds__(BLOCK) k_direct(Scene sc, Frame f, gpuart_params P, uint32_t n_slots, float4 *__restrict__ out,
                                                  uint4 *spill, unsigned long long *counters) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    WorkCounters wc = {0, 0, {0, 0, 0, 0}, 0, 0, 0};
    for (uint32_t slot = blockIdx.x * BLOCK + threadIdx.x; slot < n_slots; slot += gridDim.x * BLOCK) {
        uint32_t lx, ly;
        if (!slot_pixel(f, slot, lx, ly)) continue;
        F3 rs, rd;
        camera_ray(f, f.x0 + lx, frame_y(f, ly), rs, rd);
        F3 c = direct_lighting_pixel<REFWORK>(sc, P, rs, rd, st, &wc);
        out[(size_t)ly * f.tw + lx] = make_float4(c.x, c.y, c.z, 1.0f);
    }
    if (REFWORK) flush_counters(wc, 0, counters);
}

// ---- direct lighting with persistent lanes ------------------------------------------------------------
// One kernel, as k_direct, but a lane does not wait for its wave: it carries one pixel through the queries of
// direct_lighting.glsl:134-207 (primary ray, at most one mirror bounce off a specular user sphere, Sun-shadow query,
// query towards an emissive user sphere) as a small state machine and takes the next pixel from a shared cursor when it
// is done, so that all 64 lanes keep traversing (the loop of k_trace). The arithmetic per pixel is that of
// direct_lighting_pixel, operation for operation; the terms that wait for a visibility answer are computed before the
// query is issued and added, in the shader's order, when the answer arrives. Single kernel: no chain of dependent
// launches, which is what the interactive mode needs (one frame, then display).
enum { DL_PRIMARY = 0, DL_SUN = 1, DL_EM = 2 };

#ifndef GD_DIRECT_WAVES
#define GD_DIRECT_WAVES 4  // waves per SIMD: the launch runs alone with 16 waves per CU, so 128 VGPRs are free to use (at 5
                           // waves / 96 VGPRs the pixel's pending terms spilled to scratch)
#endif
template <int TYPES>
__global__ void __launch_bounds__(BLOCK, GD_DIRECT_WAVES) k_direct_persistent(Scene sc, Frame f, gpuart_params P, uint32_t n_slots,
                                                                          float4 *__restrict__ out, uint4 *spill, uint32_t *cursor,
                                                                          TraceTuning tune) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    // thin-wave modes (device_scene.h) for the end of the frame: once the pixel cursor is dry and the wave is down to 32 (16)
    // pixels, pairs (quads) of lanes carry them
    constexpr bool THIN_OK = GD_TRACE_THIN > 1 && GD_BOXES_OF(TYPES) == GD_BOXES_FAST;
    // Nearer child first (device_scene.h, GD_NEAREST) does not pay here: a frame's primary rays are coherent and half of its queries are
    // Sun-shadow queries, which keep the reference's order anyway — with the certificate's bookkeeping the frame took 0.710 instead of
    // 0.688 ms at 1080p, 1.715 instead of 1.695 at 4K (profiles/r04/direct_lighting_order.txt). GD_NEAREST_DIRECT=1 turns it on.
    constexpr bool NEAR = GD_NEAREST_OF(TYPES) && GD_NEAREST_DIRECT;
    uint32_t M = 1, sub = 0;  // M wave-uniform
    const float AMBIENT = 0.15f;
    const F3 sun = f3(P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]);
    const F3 usc = f3(P.userSphere[0], P.userSphere[1], P.userSphere[2]);
    const uint32_t n = n_slots;
    const uint32_t static_end = gridDim.x * tune.chunk;
    uint32_t chunk_next = min(blockIdx.x * tune.chunk, n), chunk_end = min((blockIdx.x + 1) * tune.chunk, n);  // wave-uniform
    bool exhausted = false;                                                                                   // wave-uniform
    uint32_t pixel = SLOT_INVALID;          // index into `out` of the pixel this lane works on
    int stage = DL_PRIMARY, bounce = 0;
    bool refwalk = false;                   // this lane's query is on its second walk, in the reference's order (trav_settle)
    F3 ro = f3(0, 0, 0), rd = f3(1, 0, 0), rdiv = f3(1, 1, 1);
    F3 cw = f3(1, 1, 1), acc = f3(0, 0, 0), sun_term = f3(0, 0, 0), em_term = f3(0, 0, 0), em_dir = f3(0, 0, 0), ambient = f3(0, 0, 0);
    float em_dist = 0;
    Trav t; t.state = TRAV_DONE; t.closest = 0; t.hit_prim = GD_NO_PRIM; t.node = 0; t.entry = 0; t.second = 0;
    WorkCounters wc = {0, 0, {0, 0, 0, 0}, 0, 0, 0};

    auto start_query = [&](F3 o, F3 d) {
        ro = o; rd = d;
        refwalk = false;
        rdiv = f3(1 / d.x, 1 / d.y, 1 / d.z);
        trav_init<GD_BOXES_OF(TYPES)>(sc, Ray{ro, rd}, rdiv, t, st, &wc, false, NEAR && stage != DL_SUN);
    };

    for (;;) {
        // ---- idle lanes take the next pixels
        unsigned long long idle = __ballot(pixel == SLOT_INVALID);
        while (idle && !exhausted) {
            if (chunk_next == chunk_end) {
                if (static_end >= n) { exhausted = true; break; }
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(cursor, tune.chunk);
                base = wave_value(base) + static_end;
                if (base >= n) { exhausted = true; break; }
                chunk_next = base;
                chunk_end = min(base + tune.chunk, n);
            }
            uint32_t want = (uint32_t)__popcll(idle), take = min(want, chunk_end - chunk_next);
            uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane_id()) - 1));
            if (pixel == SLOT_INVALID && rank < take) {
                uint32_t lx, ly;
                if (slot_pixel(f, chunk_next + rank, lx, ly)) {
                    pixel = ly * f.tw + lx;
                    F3 rs0, rd0;
                    camera_ray(f, f.x0 + lx, frame_y(f, ly), rs0, rd0);
                    cw = f3(1, 1, 1); acc = f3(0, 0, 0);
                    stage = DL_PRIMARY; bounce = 0;
                    start_query(rs0, rd0);
                }
            }
            chunk_next += take;
            idle = __ballot(pixel == SLOT_INVALID);
            if (take == want) break;
        }
        const unsigned long long flying = __ballot(pixel != SLOT_INVALID && sub == 0);
        if (flying == 0) {
            if (exhausted) break;
            continue;
        }
        if (THIN_OK && exhausted && M < (uint32_t)GD_TRACE_THIN) {
            const uint32_t left = (uint32_t)__popcll(flying);
            const uint32_t to = left <= BLOCK / 4 && GD_TRACE_THIN >= 4 ? 4u : left <= BLOCK / 2 ? 2u : 1u;
            if (to > M) {
                __shared__ uint32_t xfer[BLOCK];
                uint32_t tag = (uint32_t)stage | ((uint32_t)bounce << 8) | (refwalk ? 1u << 16 : 0u);
                int src = lane_id();
                thin_regroup(to, flying, xfer, ring_a, ring_b, spill, pixel, tag, ro, rd, rdiv, t, st, &src);
                stage = (int)(tag & 255u); bounce = (int)((tag >> 8) & 255u); refwalk = (tag >> 16) != 0;
                // the pixel's pending terms travel with it
                auto move3 = [&](F3 &v) { v = f3(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64)); };
                move3(cw); move3(acc); move3(sun_term); move3(em_term); move3(em_dir); move3(ambient);
                em_dist = __shfl(em_dist, src, 64);
                if ((uint32_t)lane_id() / to >= left) pixel = SLOT_INVALID;
                M = to;
                sub = (uint32_t)lane_id() & (M - 1);
            }
        }
        if (THIN_OK && M > 1) {
            // the loop below with M lanes per pixel (the cursor is dry: pixels only ever finish)
            auto thin_rounds = [&](auto width) {
                constexpr int W = decltype(width)::value;
                constexpr unsigned long long LEAD = W == 4 ? 0x1111111111111111ull : 0x5555555555555555ull;
                for (;;) {
                    // (k_run's L1-warming loads for stacked children buy nothing here: 0.641 / 1.477 against 0.639 / 1.458 ms per 1080p / 4K frame)
                    if (t.state == TRAV_DESCEND) trav_step_box_thin<W, NEAR>(sc, ro, rd, rdiv, t, st, sub, stage != DL_SUN && !refwalk);
                    const unsigned long long at_leaf = __ballot((t.state & 1) != 0) & LEAD;
                    unsigned long long busy = __ballot(t.state != TRAV_DONE) & LEAD;
                    const uint32_t waiting = (uint32_t)__popcll(at_leaf);
                    if (at_leaf && (W * waiting >= tune.leaf_lanes || tune.leaf_share * waiting >= (uint32_t)__popcll(busy))) {
                        if (t.state & 1) {
                            trav_step_leaf_thin<W, TYPES, NEAR>(sc, ro, rd, t, st, sub, stage != DL_SUN && !refwalk);
                            if (stage == DL_SUN && t.hit_prim != GD_NO_PRIM) t.state = TRAV_DONE;  // only "anything hit?" is asked
                        }
                        busy = __ballot(t.state != TRAV_DONE) & LEAD;
                    }
                    if (!busy) break;
                    // lanes with an answer move their pixel on (the others stand still meanwhile): once they are a fair share of the wave
                    const uint32_t answered = (uint32_t)__popcll(__ballot(pixel != SLOT_INVALID && t.state == TRAV_DONE) & LEAD);
                    if (2 * answered >= (uint32_t)__popcll(busy)) break;
                }
            };
            if (M == 2) thin_rounds(std::integral_constant<int, 2>());
            else thin_rounds(std::integral_constant<int, 4>());
        } else
        // ---- traverse until enough lanes have an answer (a lane without a pixel is in state DONE)
        for (;;) {
            if (t.state == TRAV_DESCEND) trav_step_box<false, GD_BOXES_OF(TYPES), NEAR>(sc, Ray{ro, rd}, rdiv, t, st, nullptr, stage != DL_SUN && !refwalk);
            unsigned long long at_leaf = __ballot((t.state & 1) != 0);
            unsigned long long descending = __ballot(t.state == TRAV_DESCEND);
            const uint32_t waiting = (uint32_t)__popcll(at_leaf);
            if (at_leaf && (waiting >= tune.leaf_lanes || tune.leaf_share * waiting >= waiting + (uint32_t)__popcll(descending))) {
                if (t.state & 1) {
                    trav_step_leaf<false, false, TYPES, NEAR>(sc, Ray{ro, rd}, t, st, nullptr, stage != DL_SUN && !refwalk);
                    if (stage == DL_SUN && t.hit_prim != GD_NO_PRIM) t.state = TRAV_DONE;  // only "anything hit?" is asked
                }
                descending = __ballot(t.state == TRAV_DESCEND);
                at_leaf = __ballot((t.state & 1) != 0);
            }
            unsigned long long busy = descending | at_leaf;
            if (!busy) break;
            if (64u - (uint32_t)__popcll(busy) >= tune.refill_lanes) break;
        }
        // ---- a finished nearest-first query that cannot vouch for its answer walks again, in the reference's order (device_scene.h)
        if (NEAR && pixel != SLOT_INVALID && t.state == TRAV_DONE && trav_settle<NEAR>(t, stage != DL_SUN && !refwalk)) {
            refwalk = true;
            trav_init<GD_BOXES_OF(TYPES)>(sc, Ray{ro, rd}, rdiv, t, st, &wc, false);
        }
        // ---- lanes with an answer move their pixel on (direct_lighting.glsl:134-207); replicas do so identically (the one store
        //      of a finished pixel is the same value to the same address)
        if (pixel != SLOT_INVALID && t.state == TRAV_DONE) {
            bool finished = false;
            if (stage == DL_PRIMARY) {
                Surface h; bool ush;
                resolve_hit(sc, Ray{ro, rd}, t.closest, t.hit_prim, P.userSphere, h, ush);
                if ((P.userSphereFlags & 2u) && ush) {
                    if (bounce == 0) {
                        cw = cw * primitive_color(P_SPHERE);
                        bounce = 1;
                        start_query(h.p, reflect3(rd, h.n));
                    } else {
                        finished = true;  // the loop of the shader ends after its second iteration: out stays 0
                    }
                } else if ((P.userSphereFlags & 1u) && ush) {
                    acc = f3(1, 1, 1);
                    finished = true;
                } else if (h.ptype != -1) {
                    const F3 diffuse = primitive_color(h.ptype) * cw;
                    ambient = AMBIENT * diffuse;
                    sun_term = lambert(sun, h.n, diffuse);
                    if (P.userSphereFlags & 1u) {
                        const F3 dts = usc - h.p;
                        em_dist = length3(dts);
                        em_dir = f3(dts.x / em_dist, dts.y / em_dist, dts.z / em_dist);
                        const F3 l = lambert(em_dir, h.n, diffuse);
                        const float d2 = dot3(dts, dts);  // dist*dist: NIR folds sqrt(a)*sqrt(a) to |a|
                        em_term = f3(l.x / d2, l.y / d2, l.z / d2);
                    }
                    // The reference issues the Sun-shadow query unconditionally (direct_lighting.glsl:170-180); where dot(sun, n) <= 0 its
                    // Lambert term is vec3(0) whatever the query says (:95-99) and `out += vec3(0)` changes nothing, so this fast
                    // kernel skips it (k_direct, modes 1 / 2, performs every reference query)
                    if (P.sunEnabled == 1 && dot3(sun, h.n) > 0) { stage = DL_SUN; start_query(h.p, sun); }
                    else if (P.userSphereFlags & 1u) { stage = DL_EM; start_query(h.p, em_dir); }
                    else { acc = acc + ambient; finished = true; }
                } else {
                    acc = cw * sky_color(rd, P.sunDirAlt);
                    finished = true;
                }
            } else if (stage == DL_SUN) {
                if (sun_visible(P, ro, sun, t.hit_prim)) acc = acc + sun_term;
                if (P.userSphereFlags & 1u) { stage = DL_EM; start_query(ro, em_dir); }
                else { acc = acc + ambient; finished = true; }
            } else {
                if (t.hit_prim == GD_NO_PRIM || t.closest > em_dist) acc = acc + em_term;
                acc = acc + ambient;
                finished = true;
            }
            if (finished) {
                out[pixel] = make_float4(acc.x, acc.y, acc.z, 1.0f);
                pixel = SLOT_INVALID;
            }
        }
    }
}

template <bool REFWORK>
__global__ void __launch_bounds__(BLOCK) k_pt_mega(Scene sc, Frame f, gpuart_params P, float4 seed, int npaths, uint32_t n_slots,
                                                   float4 *__restrict__ accum, uint4 *spill, unsigned long long *counters) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    WorkCounters wc = {0, 0, {0, 0, 0, 0}, 0, 0, 0};
    uint32_t segments = 0;
    for (uint32_t slot = blockIdx.x * BLOCK + threadIdx.x; slot < n_slots; slot += gridDim.x * BLOCK) {
        uint32_t lx, ly;
        if (!slot_pixel(f, slot, lx, ly)) continue;
        F3 rs, rd;
        camera_ray(f, f.x0 + lx, frame_y(f, ly), rs, rd);
        F3 c = path_tracing_pixel<REFWORK>(sc, P, seed, npaths, rs, rd, st, &wc, segments);
        size_t idx = (size_t)ly * f.tw + lx;
        float4 prev = accum[idx];
        accum[idx] = make_float4(prev.x + c.x, prev.y + c.y, prev.z + c.z, prev.w);
    }
    if (REFWORK) flush_counters(wc, segments, counters);
}

__global__ void k_scale_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n, float divide_by) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float4 v = src[i];
        dst[i] = make_float4(v.x / divide_by, v.y / divide_by, v.z / divide_by, v.w);
    }
}

}  // namespace
