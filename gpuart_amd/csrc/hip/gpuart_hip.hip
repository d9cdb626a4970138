// gpuart_hip.hip — render kernels and the C-ABI launcher of libgpuart_hip.so (gfx950 only).
// See include/gpuart_hip.h for the boundary and DESIGN.md for layout / kernel notes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

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
    uint32_t leaf_lanes;    ///< primitive tests are issued once this many lanes wait at a leaf
};

// =================================================================================================
// Kernels
// =================================================================================================
namespace {

/// Per-path state of the wavefront pipeline, one slot per pixel of the tile in 8x8-tile-major order
/// (slot s: tile s/64, pixel s%64 inside it), all arrays SoA and 16-byte aligned.
struct PathBuffers {
    float4 *ray_o, *ray_d;   ///< ray of the current / next segment
    uint2 *hit;              ///< closest-hit result of the current segment: (bits(t), primitive index)
    float4 *cw;              ///< colorWeight
    float4 *pc;              ///< pathColor
    float4 *sun;             ///< pending Sun term (xyz), w = bits(1: the path ends after its shadow query)
    float4 *color;           ///< colour of the earlier paths of this pass (NumPathsPerPixel > 1)
    uint32_t *queue[2];      ///< slots that trace segment s (ping-pong)
    uint32_t *shadow_queue;  ///< slots with a pending Sun shadow query
    uint32_t *counters;      ///< per segment s: [4s] rays, [4s+1] fetch cursor, [4s+2] shadow rays, [4s+3] fetch cursor
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

GD_FN void flush_counters(const WorkCounters &wc, uint32_t segments, unsigned long long *g) {
    // one atomic per counter per wavefront
    uint32_t v[7] = {wc.rays, wc.nodes, wc.prims[0], wc.prims[1], wc.prims[2], wc.prims[3], segments};
    for (int k = 0; k < 7; k++) {
        unsigned long long s = v[k];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane_id() == 0 && s) atomicAdd(&g[k], s);
    }
}

/// slot -> pixel of the tile; false for the padding slots of ragged edge tiles.
GD_FN bool slot_pixel(const Frame &f, uint32_t slot, uint32_t &lx, uint32_t &ly) {
    uint32_t tiles_x = (f.tw + 7) / 8;
    uint32_t t = slot >> 6, w = slot & 63;
    lx = (t % tiles_x) * 8 + (w & 7);
    ly = (t / tiles_x) * 8 + (w >> 3);
    return lx < f.tw && ly < f.th;
}

GD_FN TravStack make_stack(uint2 *ring_a, float *ring_b, uint4 *spill, uint32_t total_lanes) {
    TravStack st;
    st.ring_a = ring_a + lane_id();
    st.ring_b = ring_b + lane_id();
    st.ring_stride = BLOCK;
    st.spill = spill + (size_t)blockIdx.x * BLOCK + lane_id();
    st.spill_stride = total_lanes;
    st.reset();
    return st;
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
        slot_pixel(f, slot % b.n_slots, lx, ly);
        passcolor[(size_t)(slot / b.n_slots) * b.tile_pixels + (size_t)ly * f.tw + lx] = make_float4(c.x, c.y, c.z, 0);
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
        bool valid = slot_pixel(f, slot % b.n_slots, lx, ly);
        const float4 seed = seeds.seed[slot / b.n_slots];
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
                b.cw[slot] = make_float4(1, 1, 1, 0);
                b.pc[slot] = make_float4(0, 0, 0, 0);
                q = slot;
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
#ifndef GD_TRACE_WAVES
#define GD_TRACE_WAVES 5  // waves per SIMD the register allocation must allow (<= 96 VGPRs)
#endif
#ifndef GD_TRACE_WAVES_LEAN
#define GD_TRACE_WAVES_LEAN 6  // the kernels without cone / sphere code fit 6 waves per SIMD (<= 80 VGPRs)
#endif
template <bool COUNT, int TYPES>
__global__ void __launch_bounds__(BLOCK, TYPES == GD_FLAT_TYPES ? GD_TRACE_WAVES_LEAN : GD_TRACE_WAVES)
k_trace(Scene sc, Frame f, gpuart_params P, PathBuffers b, int seg_c, int seg_s, int any_shadow, int j, int npaths, float4 *accum,
        uint4 *spill, unsigned long long *gcounters, TraceTuning tune) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    const uint32_t *queue_c = b.queue[seg_c & 1];
    const uint32_t n_c = seg_c >= 0 ? b.counters[4 * seg_c] : 0u;
    const uint32_t n = n_c + (seg_s >= 0 ? b.counters[4 * seg_s + 2] : 0u);
    uint32_t *cursor = &b.counters[seg_c >= 0 ? 4 * seg_c + 1 : 4 * seg_s + 3];
    const F3 sun = f3(P.sunDirAlt[0], P.sunDirAlt[1], P.sunDirAlt[2]);
    WorkCounters wc = {0, 0, {0, 0, 0, 0}};

    // The first chunk of every wave is static (chunk index = workgroup index); later chunks come from the
    // shared cursor, which therefore starts behind the static ones. No atomic at all for small queues.
    const uint32_t static_end = gridDim.x * tune.chunk;
    uint32_t chunk_next = min(blockIdx.x * tune.chunk, n), chunk_end = min((blockIdx.x + 1) * tune.chunk, n);  // wave-uniform
    bool exhausted = false;                                                                                   // wave-uniform
    uint32_t slot = SLOT_INVALID;
    bool shadow = false;                    // this lane's ray is a Sun-shadow query
    F3 ro = f3(0, 0, 0), rd = f3(1, 0, 0);  // the ray (two F3 locals: a long-lived Ray aggregate ends up in scratch)
    F3 rdiv = f3(1, 1, 1);
    Trav t; t.state = TRAV_DONE; t.closest = 0; t.hit_prim = GD_NO_PRIM; t.node = 0; t.entry = 0;

    for (;;) {
        // ---- refill idle lanes from the queue
        unsigned long long idle = __ballot(slot == SLOT_INVALID);
        while (idle && !exhausted) {
            if (chunk_next == chunk_end) {
                if (static_end >= n) { exhausted = true; break; }
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(cursor, tune.chunk);
                base = __shfl(base, 0, 64) + static_end;
                if (base >= n) { exhausted = true; break; }
                chunk_next = base;
                chunk_end = min(base + tune.chunk, n);
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
                    ro = xyz(b.ray_o[s]);
                    rd = sh ? sun : xyz(b.ray_d[s]);
                    rdiv = f3(1 / rd.x, 1 / rd.y, 1 / rd.z);
                    trav_init(sc, Ray{ro, rd}, rdiv, t, st, &wc, COUNT);
                }
            }
            chunk_next += take;
            idle = __ballot(slot == SLOT_INVALID);
            if (take == want) break;
        }
        if (__ballot(slot != SLOT_INVALID) == 0) {
            if (exhausted) break;
            continue;
        }
        // ---- traverse until enough lanes have finished (a lane without a ray is in state DONE)
        for (;;) {
            if (t.state == TRAV_DESCEND) trav_step_box<COUNT>(sc, Ray{ro, rd}, rdiv, t, st, COUNT ? &wc : nullptr);
            unsigned long long at_leaf = __ballot((t.state & 1) != 0);  // TRAV_LEAF = 1, TRAV_LEAF_TRIS = 3
            unsigned long long descending = __ballot(t.state == TRAV_DESCEND);
            if (at_leaf && ((uint32_t)__popcll(at_leaf) >= tune.leaf_lanes || !descending)) {
                if (t.state & 1) {
                    trav_step_leaf<false, COUNT, TYPES>(sc, Ray{ro, rd}, t, st, COUNT ? &wc : nullptr);
                    // the reference only asks a shadow query whether anything was hit: one accepted hit settles it
                    if (shadow && any_shadow && t.hit_prim != GD_NO_PRIM) t.state = TRAV_DONE;
                }
                descending = __ballot(t.state == TRAV_DESCEND);
                at_leaf = __ballot((t.state & 1) != 0);
            }
            unsigned long long busy = descending | at_leaf;
            if (!busy) break;
            if (!exhausted && 64u - (uint32_t)__popcll(busy) >= tune.refill_lanes) break;
        }
        // ---- retire finished rays
        if (slot != SLOT_INVALID && t.state == TRAV_DONE) {
            if (!shadow) {
                b.hit[slot] = make_uint2(__float_as_uint(t.closest), t.hit_prim);
            } else {
                float4 term = b.sun[slot];
                F3 pathColor = xyz(b.pc[slot]);
                if (sun_visible(P, ro, sun, t.hit_prim)) pathColor = pathColor + xyz(term);
                if (__float_as_uint(term.w) & 1u) path_commit(f, b, accum, slot, j, npaths, pathColor);
                else b.pc[slot] = make_float4(pathColor.x, pathColor.y, pathColor.z, 0);
            }
            slot = SLOT_INVALID;
        }
    }
    if (COUNT) flush_counters(wc, 0, gcounters);
}

// ---- wavefront stage 2: shade segment `seg` of every path in queue[seg&1] (path_tracing.glsl:182-233) ---
template <bool REFWORK>
__global__ void __launch_bounds__(BLOCK) k_shade(Scene sc, Frame f, gpuart_params P, SeedBatch seeds, PathBuffers b, int seg,
                                                 int j, int npaths, float4 *accum, unsigned long long *gcounters) {
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
            Ray r; r.o = xyz(b.ray_o[slot]); r.d = xyz(b.ray_d[slot]);
            uint2 h = b.hit[slot];
            F3 cw = xyz(b.cw[slot]), pathColor = xyz(b.pc[slot]);
            F3 rstart = r.o, rdir = r.d;
            segments++;
            const float4 seed = seeds.seed[slot / b.n_slots];
            ShadeResult s = path_shade(sc, P, seed, seg, r, __uint_as_float(h.x), h.y, rstart, rdir, cw, pathColor);
            if (s.broke) {
                uint32_t lx, ly; F3 rs0, rd0;
                slot_pixel(f, slot % b.n_slots, lx, ly);
                camera_ray(f, f.x0 + lx, frame_y(f, ly), rs0, rd0);
                path_commit(f, b, accum, slot, j, npaths, path_finish(P, rd0, seg, s.ush, s.specular, pathColor));
            } else {
                go_on = s.next == PATH_CONTINUES;
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
      base1 = __shfl(base1, 0, 64);
      base2 = __shfl(base2, 0, 64);
      for (uint32_t i = threadIdx.x; i < n_next; i += BLOCK) next_queue[base1 + i] = stage_next[i];
      for (uint32_t i = threadIdx.x; i < n_shadow; i += BLOCK) b.shadow_queue[base2 + i] = stage_shadow[i];
      __syncthreads();
    }
    if (REFWORK) {
        WorkCounters z = {0, 0, {0, 0, 0, 0}};
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
    WorkCounters wc = {0, 0, {0, 0, 0, 0}};
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

template <bool REFWORK>
__global__ void __launch_bounds__(BLOCK) k_pt_mega(Scene sc, Frame f, gpuart_params P, float4 seed, int npaths, uint32_t n_slots,
                                                   float4 *__restrict__ accum, uint4 *spill, unsigned long long *counters) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    WorkCounters wc = {0, 0, {0, 0, 0, 0}};
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

// ---- test-hook kernels ---------------------------------------------------------------------------
__global__ void k_test_random(const float4 *in, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = in[i];
    out[i] = make_float4(random1(v.x), random2(v.x, v.y), random3(f3(v.x, v.y, v.z)), random4(v));
}
__global__ void k_test_math(const float4 *in, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincos_lp(in[i].x, s, c);
    out[i] = make_float4(s, c, pow_lp(in[i].y, 16.0f), sqrtf(in[i].y));
}
__global__ void k_test_hemisphere(const float4 *v, const float4 *ri, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 r = random_hemisphere_direction(xyz(v[i]), xyz(ri[i]));
    out[i] = make_float4(r.x, r.y, r.z, 0);
}
__global__ void k_test_inside_cone(const float4 *v, const float4 *nrm, const float4 *ri, float ha, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 r = random_direction_inside_cone(xyz(v[i]), xyz(nrm[i]), ha, xyz(ri[i]));
    out[i] = make_float4(r.x, r.y, r.z, 0);
}
struct Float4Arg { float v[4]; };
__global__ void k_test_sky(const float4 *dir, Float4Arg sda, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 r = sky_color(xyz(dir[i]), sda.v);
    out[i] = make_float4(r.x, r.y, r.z, 0);
}
/// recs: n device-layout primitive records (3 quads each)
__global__ void k_test_intersect(const float4 *rs, const float4 *rd, const float4 *recs, int n, float4 *o0, float4 *o1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = xyz(rs[i]); r.d = xyz(rd[i]);
    float pos; F3 p = f3(0, 0, 0), nn = f3(0, 0, 0); int t;
    prim_hit(r, recs[3 * i], recs[3 * i + 1], recs[3 * i + 2], pos, p, nn, t);
    if (pos > 0) { o0[i] = make_float4(pos, p.x, p.y, p.z); o1[i] = make_float4(nn.x, nn.y, nn.z, 0); }
    else { o0[i] = make_float4(pos, 0, 0, 0); o1[i] = make_float4(0, 0, 0, 0); }
}
__global__ void k_test_aabb(const float4 *rs, const float4 *rd, const float4 *bmin, const float4 *bmax, int n, float4 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = xyz(rs[i]); r.d = xyz(rd[i]);
    float pos;
    bool h = aabb_entry(r, f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z), xyz(bmin[i]), xyz(bmax[i]), pos);
    out[i] = make_float4(h ? 1.0f : 0.0f, h ? pos : 0.0f, 0, 0);
}
template <bool ANY>
__global__ void __launch_bounds__(BLOCK) k_test_traverse(Scene sc, const float4 *rs, const float4 *rd, Float4Arg us, int n,
                                                         float4 *o0, float4 *o1, uint4 *spill) {
    __shared__ uint2 ring_a[GD_RING * BLOCK];
    __shared__ float ring_b[GD_RING * BLOCK];
    TravStack st = make_stack(ring_a, ring_b, spill, gridDim.x * BLOCK);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        Ray r; r.o = xyz(rs[i]); r.d = xyz(rd[i]);
        float closest; uint32_t prim;
        traverse<ANY, false>(sc, r, st, closest, prim, nullptr);
        if (ANY) {
            o0[i] = make_float4(prim != GD_NO_PRIM ? 1.0f : 0.0f, 0, 0, 0);
            o1[i] = make_float4(0, 0, 0, 0);
            continue;
        }
        Surface h; h.p = f3(0, 0, 0); h.n = f3(0, 0, 0);
        bool ush;
        resolve_hit(sc, r, closest, prim, us.v, h, ush);
        if (h.ptype >= 0) {
            o0[i] = make_float4(h.pos, h.p.x, h.p.y, h.p.z);
            o1[i] = make_float4(h.n.x, h.n.y, h.n.z, (float)h.ptype + (ush ? 0.5f : 0.0f));
        } else {
            o0[i] = make_float4(-1, 0, 0, 0);
            o1[i] = make_float4(0, 0, 0, -1);
        }
    }
}
__global__ void k_test_cam_rays(Frame f, float4 *rstart, float4 *rdir) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f.tw * f.th) return;
    uint32_t lx = i % f.tw, ly = i / f.tw;
    F3 s, d;
    camera_ray(f, f.x0 + lx, frame_y(f, ly), s, d);
    rstart[i] = make_float4(s.x, s.y, s.z, 0);
    rdir[i] = make_float4(d.x, d.y, d.z, 0);
}

}  // namespace

// =================================================================================================
// Host side: context, upload, launches
// =================================================================================================
namespace {

thread_local std::string g_last_error;

// Every pipeline run in flight has its own stream (+ the primary stream). ROCm maps HIP streams onto GPU_MAX_HW_QUEUES
// hardware queues (default 4) and kernels of streams that share a queue serialise, which would undo the overlap
// the pass lanes exist for; ask for more queues unless the user has chosen a value. Must happen before the HIP
// runtime initialises, hence a load-time constructor (bench.py also sets it before importing torch).
__attribute__((constructor)) void request_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "32", 0); }

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(GPUART_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// llvmpipe's plane equation of one interpolated attribute over one triangle (DESIGN.md "UV").
void plane_coef(float x0, float y0, float x1, float y1, float x2, float y2, float a0, float a1, float a2, float c[3]) {
    float x0c = x0 - 0.5f, y0c = y0 - 0.5f;
    float dx01 = x0 - x1, dy01 = y0 - y1, dx20 = x2 - x0, dy20 = y2 - y0;
    float e = dx01 * dy20, f = dy01 * dx20;
    float ooa = 1.0f / (e - f);
    float dy20o = dy20 * ooa, dy01o = dy01 * ooa, dx20o = dx20 * ooa, dx01o = dx01 * ooa;
    float da01 = a0 - a1, da20 = a2 - a0;
    float dadx = da01 * dy20o - da20 * dy01o;
    float dady = da20 * dx01o - da01 * dx20o;
    c[0] = a0 - (dadx * x0c + dady * y0c);
    c[1] = dadx;
    c[2] = dady;
}

struct TimedLaunch {
    hipEvent_t start, stop;
    int cls;  ///< 0: a whole render call (direct frame / path-tracing pass), 1: one BVH-query kernel
};

}  // namespace

/// Everything one run of the pipeline (a group of path-tracing passes) needs while it is in flight. Several runs are in
/// flight at once, each on its own stream; their kernels fill each other's tails.
struct PassLane {
    hipStream_t main = nullptr;    ///< every kernel of the run, in order
    PathBuffers pb{};              ///< wavefront path state (passes of the run x tile slots)
    void *pathmem = nullptr;
    float4 *passcolor = nullptr;   ///< colour per pass and pixel, added to the accumulator by k_accumulate
    uint4 *spill_main = nullptr;   ///< traversal-stack overflow of the lane's BVH-query launches
    uint32_t counter_segments = 0; ///< pb.counters holds 4*(counter_segments+1) words
    hipEvent_t ev_done = nullptr;  ///< the run has finished (main stream)
    hipEvent_t ev_free = nullptr;  ///< its colour has been accumulated (primary stream): the lane may be reused
    bool used = false;
};

struct gpuart_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  ///< primary stream: accumulation (in pass order), direct lighting, copies, test hooks
    Frame frame{};
    bool have_camera = false, have_scene = false;
    float4 *d_recs = nullptr, *d_prims = nullptr;
    uint4 *d_spill = nullptr;      ///< [spill_levels][grid_lanes] traversal-stack overflow for kernels on the primary stream
    std::vector<PassLane> lanes;
    uint32_t next_lane = 0;
    uint32_t spill_levels = 0;
    uint32_t num_cus = 256;
    uint32_t grid_waves = 4096;    ///< persistent grid: one wave per block
    TraceTuning tune{128, 16, 16};
    uint32_t n_slots = 0;          ///< path slots of the tile (8x8-tile padded)
    uint32_t max_batch = 1;        ///< most passes one run of the pipeline may hold (batch_paths / tile slots, <= MAX_BATCH)
    uint32_t batch_limit = MAX_BATCH;  ///< user cap (GPUART_HIP_MAX_BATCH)
    size_t batch_paths = (size_t)16 << 20;  ///< passes are batched while one pipeline run stays within this many paths
    size_t min_run_paths = (size_t)2 << 20;  ///< a pipeline run is not made smaller than this many paths
    double plan_run_factor = 0.75; ///< run length = this x sqrt(planned work), in units of 2M paths (plan_runs)
    bool lean_kernels = true;      ///< use the BVH-query kernels specialised for the primitive types present
    uint32_t planned_passes = 0;   ///< gpuart_hip_pt_plan hint (0: unknown)
    size_t run_passes = 1;         ///< passes per pipeline run (see plan_runs)
    uint32_t lanes_in_use = 1;     ///< pass lanes cycled through (all of them unless their path state would exceed lane_budget)
    size_t lane_budget = (size_t)16 << 30;  ///< bytes of wavefront path state over all lanes
    // passes requested through gpuart_hip_pt_pass but not launched yet (same params, one seed each)
    std::vector<float4> pend_seeds;
    gpuart_params pend_params{};
    int pend_npaths = 0;
    uint64_t n_nodes = 0, n_prims = 0, scene_bytes = 0;
    uint32_t type_mask = 0;  ///< bit t set: the scene holds primitives of type t
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    uint32_t root_ref = 0;
    uint32_t max_depth = 0;
    float4 *d_direct = nullptr, *d_accum = nullptr;
    size_t tile_pixels = 0;
    unsigned long long *d_counters = nullptr;
    int mode = 0;  ///< 0 wavefront (fast), 1 reference-work (wavefront, full queries, counters), 2 megakernel
    std::vector<TimedLaunch> pending, free_events;
    double timed_ms[2] = {0, 0};
    uint64_t timed_launches[2] = {0, 0};
    int timing_level = 1;  ///< 0 none, 1 per render call, 2 also per BVH-query kernel
    float *d_scratch = nullptr;  // test hooks
    size_t scratch_bytes = 0;
};

namespace {

/// Waits for everything this context has enqueued, on all of its streams.
int drain(gpuart_hip_ctx *c) {
    for (auto &l : c->lanes) {
        if (l.main) HIP_TRY(hipStreamSynchronize(l.main));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace
extern "C" int gpuart_hip_flush(gpuart_hip_ctx *c);
namespace {

/// Passes per pipeline run. Runs should be long (a persistent launch that takes many rays per lane wastes less of its
/// instructions on draining its last rays) and numerous (their kernels fill each other's tails, and more runs than lanes
/// keeps the lanes out of step). For a planned sequence of u units of work (1 unit = 2M paths, one 1080p pass) the best
/// run length measured on cfg3 was 1, 1, 2-3, 3-4, 6 units for u = 2, 4, 8, 20, 64 — about 0.75 sqrt(u); never below
/// `min_run_paths`, never above max_batch (16M paths). Without a plan: 8M paths.
void plan_runs(gpuart_hip_ctx *c) {
    if (!c->n_slots) { c->run_passes = 1; return; }
    const double unit = (double)((size_t)2 << 20);
    const size_t min_run = std::max<size_t>(1, c->min_run_paths / c->n_slots);
    size_t want;
    if (c->planned_passes) {
        const double u = (double)c->planned_passes * c->n_slots / unit;
        want = (size_t)(c->plan_run_factor * std::sqrt(u) * unit / c->n_slots);
    } else {
        want = std::max<size_t>(1, ((size_t)8 << 20) / c->n_slots);
    }
    c->run_passes = std::min<size_t>(c->max_batch, std::max(min_run, want));
}

int realloc_tile(gpuart_hip_ctx *c) {
    int r = gpuart_hip_flush(c);
    if (r) return r;
    if ((r = drain(c))) return r;
    if (c->d_direct) { (void)hipFree(c->d_direct); c->d_direct = nullptr; }
    if (c->d_accum) { (void)hipFree(c->d_accum); c->d_accum = nullptr; }
    c->tile_pixels = (size_t)c->frame.tw * c->frame.th;
    if (!c->tile_pixels) return 0;
    HIP_TRY(hipMalloc(&c->d_direct, c->tile_pixels * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->d_accum, c->tile_pixels * sizeof(float4)));
    HIP_TRY(hipMemsetAsync(c->d_direct, 0, c->tile_pixels * sizeof(float4), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_accum, 0, c->tile_pixels * sizeof(float4), c->stream));
    // wavefront path state: one slot per pixel of the 8x8-tile-padded tile, per pass in flight
    const size_t tiles = (size_t)((c->frame.tw + 7) / 8) * ((c->frame.th + 7) / 8);
    const size_t n = tiles * 64;
    if (n > 0xfffffff0ull) return fail(GPUART_HIP_ERR_ARG, "tile too large");
    c->n_slots = (uint32_t)n;
    // Several passes run through the pipeline together (slot = pass x pixel) while that stays within `batch_paths`
    // paths: a persistent k_trace wave then takes many rays per lane, and the drain at the end of every launch — waves
    // finishing their last, long rays with few lanes busy — shrinks relative to the useful work (SQ_INSTS_VALU per
    // ray falls by a quarter from 2M to 16M paths per launch at 1080p).
    size_t B = std::max<size_t>(1, std::min<size_t>(c->batch_limit, c->batch_paths / n));
    if (n * B > 0x7ffffff0ull) return fail(GPUART_HIP_ERR_ARG, "tile too large");  // two queues share one 32-bit index space in k_trace
    for (auto &l : c->lanes) {
        if (l.pathmem) { (void)hipFree(l.pathmem); l.pathmem = nullptr; }
        l.used = false;
    }
    c->next_lane = 0;
    // Lanes within the memory budget; when the device cannot give that much (other tenants), fewer lanes and then
    // smaller runs are tried before giving up — results do not depend on either.
    size_t lanes = 0, bytes = 0;
    for (;;) {
        bytes = n * B * (6 * sizeof(float4) + sizeof(uint2) + 3 * sizeof(uint32_t)) + B * c->tile_pixels * sizeof(float4);
        if (!lanes) lanes = std::min<size_t>(c->lanes.size(), std::max<size_t>(2, c->lane_budget / bytes));
        size_t got = 0;
        while (got < lanes && hipMalloc(&c->lanes[got].pathmem, bytes) == hipSuccess) got++;
        if (got == lanes) break;
        (void)hipGetLastError();  // clear the out-of-memory error
        for (size_t li = 0; li < got; li++) { (void)hipFree(c->lanes[li].pathmem); c->lanes[li].pathmem = nullptr; }
        if (lanes > 2) lanes = std::max<size_t>(2, lanes / 2);
        else if (B > 1) { B = (B + 1) / 2; lanes = 0; }
        else if (lanes > 1) lanes = 1;
        else return fail(GPUART_HIP_ERR_DEVICE, "out of device memory for the path state of one pass");
    }
    c->max_batch = (uint32_t)B;
    c->lanes_in_use = (uint32_t)lanes;
    plan_runs(c);
    for (size_t li = 0; li < lanes; li++) {
        PassLane &l = c->lanes[li];
        char *m = (char *)l.pathmem;
        PathBuffers &b = l.pb;
        const size_t nb = n * B;
        b.ray_o = (float4 *)m; m += nb * sizeof(float4);
        b.ray_d = (float4 *)m; m += nb * sizeof(float4);
        b.cw = (float4 *)m; m += nb * sizeof(float4);
        b.pc = (float4 *)m; m += nb * sizeof(float4);
        b.sun = (float4 *)m; m += nb * sizeof(float4);
        b.color = (float4 *)m; m += nb * sizeof(float4);
        l.passcolor = (float4 *)m; m += B * c->tile_pixels * sizeof(float4);
        b.hit = (uint2 *)m; m += nb * sizeof(uint2);
        b.queue[0] = (uint32_t *)m; m += nb * sizeof(uint32_t);
        b.queue[1] = (uint32_t *)m; m += nb * sizeof(uint32_t);
        b.shadow_queue = (uint32_t *)m;
        b.n_slots = (uint32_t)n;
        b.batch = 1;
        b.tile_pixels = (uint32_t)c->tile_pixels;
    }
    return 0;
}

int ensure_segment_counters(gpuart_hip_ctx *c, PassLane &l, uint32_t nseg) {
    if (l.pb.counters && l.counter_segments >= nseg) return 0;
    if (l.pb.counters) { int r = drain(c); if (r) return r; (void)hipFree(l.pb.counters); l.pb.counters = nullptr; }
    HIP_TRY(hipMalloc(&l.pb.counters, 4 * ((size_t)nseg + 1) * sizeof(uint32_t)));
    l.counter_segments = nseg;
    return 0;
}

int ensure_spill(gpuart_hip_ctx *c) {
    const uint32_t levels = c->max_depth > GD_RING ? c->max_depth - GD_RING : 0;
    if (c->d_spill && c->spill_levels >= levels) return 0;
    int r = drain(c);
    if (r) return r;
    const size_t bytes = ((size_t)levels + 1) * c->grid_waves * BLOCK * sizeof(uint4);
    if (c->d_spill) { (void)hipFree(c->d_spill); c->d_spill = nullptr; }
    HIP_TRY(hipMalloc(&c->d_spill, bytes));
    for (auto &l : c->lanes) {
        if (l.spill_main) { (void)hipFree(l.spill_main); l.spill_main = nullptr; }
        HIP_TRY(hipMalloc(&l.spill_main, bytes));
    }
    c->spill_levels = levels;
    return 0;
}

void update_uv(gpuart_hip_ctx *c) {
    float w = (float)c->frame.W, h = (float)c->frame.H;
    float *k = c->frame.uv_coef;
    plane_coef(w, 0, 0, 0, w, h, 1, 0, 1, k + 0);  // A.u (V1,V0,V2)
    plane_coef(w, 0, 0, 0, w, h, 0, 0, 1, k + 3);  // A.v
    plane_coef(w, h, 0, 0, 0, h, 1, 0, 0, k + 6);  // B.u (V2,V0,V3)
    plane_coef(w, h, 0, 0, 0, h, 1, 0, 1, k + 9);  // B.v
}

Scene scene_of(const gpuart_hip_ctx *c) {
    Scene s;
    s.recs = c->d_recs;
    memcpy(s.root_min, c->root_min, 12); memcpy(s.root_max, c->root_max, 12);
    s.root_ref = c->root_ref;
    s.prims = c->d_prims;
    return s;
}


int fold_timings(gpuart_hip_ctx *c) {
    for (auto &t : c->pending) {
        HIP_TRY(hipEventSynchronize(t.stop));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, t.start, t.stop));
        c->timed_ms[t.cls] += ms;
        c->timed_launches[t.cls]++;
        c->free_events.push_back(t);
    }
    c->pending.clear();
    return 0;
}

int begin_timed(gpuart_hip_ctx *c, TimedLaunch &t, int cls, hipStream_t stream = nullptr) {
    if (!stream) stream = c->stream;
    t.cls = cls;
    t.start = t.stop = nullptr;
    if (c->timing_level < 1) return 0;
    if (c->pending.size() >= 4096) { int r = fold_timings(c); if (r) return r; }
    if (!c->free_events.empty()) { t = c->free_events.back(); c->free_events.pop_back(); t.cls = cls; }
    else { HIP_TRY(hipEventCreate(&t.start)); HIP_TRY(hipEventCreate(&t.stop)); }
    HIP_TRY(hipEventRecord(t.start, stream));
    return 0;
}
int end_timed(gpuart_hip_ctx *c, TimedLaunch &t, hipStream_t stream = nullptr) {
    if (!t.start) return 0;
    if (!stream) stream = c->stream;
    HIP_TRY(hipEventRecord(t.stop, stream));
    c->pending.push_back(t);
    return 0;
}

// ---- canonical tree -> device layout ---------------------------------------------------------------
struct Converter {
    const float *q;
    size_t nq;
    std::vector<float4> recs, prims;
    size_t num_nodes = 0;
    uint32_t type_mask = 0;
    uint32_t max_depth = 0;
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

    /// A box that is inverted (min > max on some axis: e.g. a sphere with a negative radius, or the empty scene)
    /// or holds a NaN can never be hit by the reference's comparisons. The device's box test assumes min <= max,
    /// so such a box is replaced by a point box far outside anything a ray can reach (its entry parameter
    /// would exceed the initial `closest` of 1e19, so it is never entered).
    static void sanitize(Child &c) {
        bool ok = true;
        for (int k = 0; k < 3; k++) ok = ok && (c.bmin[k] <= c.bmax[k]);  // false for NaN too
        if (!ok)
            for (int k = 0; k < 3; k++) c.bmin[k] = c.bmax[k] = 3.0e+38f;
    }

    /// Converts the subtree at quad address `addr`. Interior nodes get a 64-byte record (pre-order, so an
    /// interior lower child's record directly follows its parent's); leaves append their primitives.
    bool node(size_t addr, uint32_t depth, Child &out) {
        if (addr + 3 > nq) { err = "node address out of range"; return false; }
        if (depth > 1024) { err = "tree deeper than 1024 levels"; return false; }
        if (depth > max_depth) max_depth = depth;
        num_nodes++;
        const float *b = q + 4 * addr;
        for (int k = 0; k < 3; k++) { out.bmin[k] = b[k]; out.bmax[k] = b[4 + k]; }
        uint32_t flags = bits(b[8]);
        if (flags & 0x80000000u) {
            uint32_t n = flags & ~0xE0000000u;
            uint32_t first = (uint32_t)(prims.size() / 3);
            if (first >= 0x3ffffff0u) { err = "too many primitives"; return false; }
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
                float4 rec[3];
                pack_prim(type, q + 4 * (a + 1), rec);
                if (i == 0) rec[0].w = fbits(type | (n << 2));  // the first primitive carries the leaf's count
                prims.push_back(rec[0]); prims.push_back(rec[1]); prims.push_back(rec[2]);
                a += 1 + LEN[type];
            }
            if (n == 0) {  // empty leaf (empty scene): one dummy record with count 0
                prims.push_back(make_float4(0, 0, 0, fbits(0))); prims.push_back(make_float4(0, 0, 0, 0)); prims.push_back(make_float4(0, 0, 0, 0));
            }
            out.ref = GD_REF_LEAF | (all_tris ? GD_REF_TRIS : 0u) | first;
            return true;
        }
        uint32_t lo = bits(b[9]), hi = bits(b[10]);
        if (lo != addr + 3) { err = "lower child does not follow its parent"; return false; }
        if (hi <= lo || hi >= nq) { err = "upper child address out of range"; return false; }
        const size_t r = recs.size() / 4;
        if (r >= 0x3ffffff0u) { err = "too many nodes"; return false; }
        recs.resize(recs.size() + 4);
        Child L, H;
        if (!node(lo, depth + 1, L)) return false;
        if (!node(hi, depth + 1, H)) return false;
        sanitize(L); sanitize(H);
        recs[4 * r + 0] = make_float4(L.bmin[0], L.bmin[1], L.bmin[2], fbits(L.ref));
        recs[4 * r + 1] = make_float4(L.bmax[0], L.bmax[1], L.bmax[2], fbits(H.ref));
        recs[4 * r + 2] = make_float4(H.bmin[0], H.bmin[1], H.bmin[2], 0);
        recs[4 * r + 3] = make_float4(H.bmax[0], H.bmax[1], H.bmax[2], 0);
        out.ref = (uint32_t)r;
        return true;
    }
};

template <class T>
int upload_vec(gpuart_hip_ctx *c, T *&dst, const std::vector<T> &v) {
    if (dst) { (void)hipFree(dst); dst = nullptr; }
    size_t bytes = (v.size() + 4) * sizeof(T);  // a little slack past the end
    HIP_TRY(hipMalloc(&dst, bytes));
    HIP_TRY(hipMemsetAsync(dst, 0, bytes, c->stream));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int ensure_scratch(gpuart_hip_ctx *c, size_t bytes) {
    if (c->scratch_bytes >= bytes) return 0;
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&c->d_scratch, bytes));
    c->scratch_bytes = bytes;
    return 0;
}

/// Runs a test kernel: copies `nin` (n x 4 float) inputs up, launches, copies `nout` outputs back.
template <class Launch>
int run_hook(gpuart_hip_ctx *c, int n, const float *const *ins, int nin, float *const *outs, int nout, Launch launch) {
    if (!c || n < 0) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    size_t one = (size_t)n * 16;
    int r = ensure_scratch(c, one * (nin + nout));
    if (r) return r;
    float4 *base = (float4 *)c->d_scratch;
    std::vector<float4 *> di, dout;
    for (int k = 0; k < nin; k++) {
        di.push_back(base + (size_t)k * n);
        HIP_TRY(hipMemcpyAsync(di[k], ins[k], one, hipMemcpyHostToDevice, c->stream));
    }
    for (int k = 0; k < nout; k++) dout.push_back(base + (size_t)(nin + k) * n);
    launch(di, dout);
    HIP_TRY(hipGetLastError());
    for (int k = 0; k < nout; k++) HIP_TRY(hipMemcpyAsync(outs[k], dout[k], one, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace

extern "C" {

const char *gpuart_hip_last_error(void) { return g_last_error.c_str(); }

int gpuart_hip_destroy(gpuart_hip_ctx *c);

int gpuart_hip_create(int device, gpuart_hip_ctx **out) {
    if (!out) return fail(GPUART_HIP_ERR_ARG, "out == NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(GPUART_HIP_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= count) return fail(GPUART_HIP_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GPUART_HIP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    gpuart_hip_ctx *c = new (std::nothrow) gpuart_hip_ctx();
    if (!c) return fail(GPUART_HIP_ERR_DEVICE, "out of host memory");
    c->device = device;
    auto env_u32 = [](const char *name, uint32_t dflt, uint32_t lo, uint32_t hi) {
        const char *v = getenv(name);
        if (!v) return dflt;
        long x = strtol(v, nullptr, 10);
        return (uint32_t)std::min<long>(hi, std::max<long>(lo, x));
    };
    c->num_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    c->grid_waves = c->num_cus * env_u32("GPUART_HIP_WAVES_PER_CU", 8, 1, 32);  // persistent grids of one-wave workgroups
    c->tune.chunk = env_u32("GPUART_HIP_CHUNK", 128, 16, 4096);
    c->tune.refill_lanes = env_u32("GPUART_HIP_REFILL_LANES", 16, 1, 64);
    c->tune.leaf_lanes = env_u32("GPUART_HIP_LEAF_LANES", 16, 1, 64);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(GPUART_HIP_ERR_DEVICE, "hipStreamCreate failed"); }
    c->lanes.resize(env_u32("GPUART_HIP_PASSES_IN_FLIGHT", 8, 1, 32));
    c->batch_limit = env_u32("GPUART_HIP_MAX_BATCH", MAX_BATCH, 1, MAX_BATCH);
    c->batch_paths = (size_t)env_u32("GPUART_HIP_BATCH_MPATHS", 16, 1, 256) << 20;
    c->plan_run_factor = env_u32("GPUART_HIP_PLAN_RUN_PERCENT", 75, 1, 1000) / 100.0;
    c->lean_kernels = env_u32("GPUART_HIP_LEAN_KERNELS", 1, 0, 1) != 0;
    c->min_run_paths = (size_t)env_u32("GPUART_HIP_MIN_RUN_KPATHS", 2048, 64, 65536) << 10;
    c->lane_budget = (size_t)env_u32("GPUART_HIP_LANE_BUDGET_MB", 16384, 64, 262144) << 20;
    for (auto &l : c->lanes) {
        if (hipStreamCreateWithFlags(&l.main, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&l.ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&l.ev_free, hipEventDisableTiming) != hipSuccess) {
            gpuart_hip_destroy(c);
            return fail(GPUART_HIP_ERR_DEVICE, "stream / event creation failed");
        }
    }
    if (hipMalloc(&c->d_counters, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream) != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c; return fail(GPUART_HIP_ERR_DEVICE, "counter allocation failed");
    }
    *out = c;
    return 0;
}

int gpuart_hip_destroy(gpuart_hip_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    c->pend_seeds.clear();
    (void)drain(c);
    for (auto &t : c->pending) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto &t : c->free_events) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto &l : c->lanes) {
        if (l.ev_done) (void)hipEventDestroy(l.ev_done);
        if (l.ev_free) (void)hipEventDestroy(l.ev_free);
        void *lp[] = {l.pathmem, l.spill_main, l.pb.counters};
        for (void *p : lp) if (p) (void)hipFree(p);
        if (l.main) (void)hipStreamDestroy(l.main);
    }
    void *ptrs[] = {c->d_recs, c->d_prims, c->d_spill, c->d_direct, c->d_accum, c->d_counters, c->d_scratch};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int gpuart_hip_resize(gpuart_hip_ctx *c, uint32_t width, uint32_t height) {
    if (!c || width == 0 || height == 0 || width > 65536 || height > 65536) return fail(GPUART_HIP_ERR_ARG, "bad frame size");
    HIP_TRY(hipSetDevice(c->device));
    c->frame.W = width; c->frame.H = height;
    c->frame.x0 = 0; c->frame.y0 = 0; c->frame.tw = width; c->frame.th = height;
    c->frame.band_rows = height; c->frame.band_stride = height;
    update_uv(c);
    return realloc_tile(c);
}

int gpuart_hip_set_tile(gpuart_hip_ctx *c, uint32_t x0, uint32_t y0, uint32_t tw, uint32_t th) {
    if (!c || !c->frame.W) return fail(GPUART_HIP_ERR_ARG, "set_tile before resize");
    if (tw == 0 || th == 0 || (uint64_t)x0 + tw > c->frame.W || (uint64_t)y0 + th > c->frame.H)
        return fail(GPUART_HIP_ERR_ARG, "tile outside the frame");
    HIP_TRY(hipSetDevice(c->device));
    c->frame.x0 = x0; c->frame.y0 = y0; c->frame.tw = tw; c->frame.th = th;
    c->frame.band_rows = th; c->frame.band_stride = th;
    return realloc_tile(c);
}

int gpuart_hip_set_tile_interleaved(gpuart_hip_ctx *c, uint32_t x0, uint32_t y0, uint32_t tw, uint32_t th_local,
                                    uint32_t band_rows, uint32_t band_stride) {
    if (!c || !c->frame.W) return fail(GPUART_HIP_ERR_ARG, "set_tile before resize");
    if (tw == 0 || th_local == 0 || band_rows == 0 || band_stride < band_rows || (uint64_t)x0 + tw > c->frame.W)
        return fail(GPUART_HIP_ERR_ARG, "bad interleaved tile");
    // the last local row must still lie inside the frame
    const uint64_t last = (uint64_t)y0 + (uint64_t)((th_local - 1) / band_rows) * band_stride + (th_local - 1) % band_rows;
    if (last >= c->frame.H) return fail(GPUART_HIP_ERR_ARG, "interleaved tile outside the frame");
    HIP_TRY(hipSetDevice(c->device));
    c->frame.x0 = x0; c->frame.y0 = y0; c->frame.tw = tw; c->frame.th = th_local;
    c->frame.band_rows = band_rows; c->frame.band_stride = band_stride;
    return realloc_tile(c);
}

int gpuart_hip_upload_bvh(gpuart_hip_ctx *c, const float *quads, size_t nquads) {
    if (!c || !quads || nquads < 3 || nquads > (1ull << 29)) return fail(GPUART_HIP_ERR_ARG, "bad tree");
    HIP_TRY(hipSetDevice(c->device));
    Converter cv;
    cv.q = quads; cv.nq = nquads;
    Converter::Child root;
    if (!cv.node(0, 0, root)) return fail(GPUART_HIP_ERR_ARG, "malformed compiled BVH: " + cv.err);
    Converter::sanitize(root);
    int r;
    if ((r = gpuart_hip_flush(c))) return r;
    if ((r = drain(c))) return r;
    if ((r = upload_vec(c, c->d_recs, cv.recs))) return r;
    if ((r = upload_vec(c, c->d_prims, cv.prims))) return r;
    memcpy(c->root_min, root.bmin, 12); memcpy(c->root_max, root.bmax, 12);
    c->root_ref = root.ref;
    c->type_mask = cv.type_mask;
    c->n_nodes = cv.num_nodes;
    c->n_prims = cv.prims.size() / 3;
    c->max_depth = cv.max_depth;
    c->scene_bytes = cv.recs.size() * 16 + cv.prims.size() * 16;
    if ((r = ensure_spill(c))) return r;
    c->have_scene = true;
    return 0;
}

int gpuart_hip_set_camera(gpuart_hip_ctx *c, const float pos[3], const float bl[3], const float dh[3], const float dv[3]) {
    if (!c || !pos || !bl || !dh || !dv) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }  // batched passes were requested with the old camera
    memcpy(c->frame.cam_pos, pos, 12); memcpy(c->frame.bottom_left, bl, 12);
    memcpy(c->frame.delta_horz, dh, 12); memcpy(c->frame.delta_vert, dv, 12);
    c->have_camera = true;
    return 0;
}

static int check_ready(gpuart_hip_ctx *c, const gpuart_params *p) {
    if (!c || !p) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    if (!c->have_scene) return fail(GPUART_HIP_ERR_ARG, "no scene uploaded");
    if (!c->have_camera) return fail(GPUART_HIP_ERR_ARG, "no camera set");
    if (!c->tile_pixels) return fail(GPUART_HIP_ERR_ARG, "no frame size set");
    return 0;
}

int gpuart_hip_render_direct(gpuart_hip_ctx *c, const gpuart_params *p) {
    int r = check_ready(c, p);
    if (r) return r;
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    HIP_TRY(hipSetDevice(c->device));
    dim3 grid(std::min<uint32_t>(c->grid_waves, c->n_slots / BLOCK));
    Scene sc = scene_of(c);
    TimedLaunch t;
    if ((r = begin_timed(c, t, 0))) return r;
    if (c->mode == 1) k_direct<true><<<grid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, c->n_slots, c->d_direct, c->d_spill, c->d_counters);
    else k_direct<false><<<grid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, c->n_slots, c->d_direct, c->d_spill, c->d_counters);
    HIP_TRY(hipGetLastError());
    return end_timed(c, t);
}

int gpuart_hip_pt_plan(gpuart_hip_ctx *c, uint32_t passes) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    c->planned_passes = passes;
    plan_runs(c);
    return 0;
}

int gpuart_hip_pt_reset(gpuart_hip_ctx *c) {
    if (!c || !c->tile_pixels) return fail(GPUART_HIP_ERR_ARG, "no frame size set");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    // passes still in flight belong to the accumulation that is being discarded: let them finish first
    int r = drain(c);
    if (r) return r;
    HIP_TRY(hipMemsetAsync(c->d_accum, 0, c->tile_pixels * sizeof(float4), c->stream));
    return 0;
}

/// Upper bound on the number of segments any path can have. colorWeight is multiplied per segment by the albedo
/// of the primitive type that was hit (path_tracing.glsl:123-126,204) and the loop stops as soon as ANY channel
/// is <= minWeight, so channel c survives at most as long as (largest albedo.c among the primitive types present
/// in the scene)^n > minWeight; the bound is the smallest such n over the channels (+1 when a product comes
/// within 1e-4 of minWeight, where fp32 rounding of a mixed product could differ). Kernels for segments beyond
/// the bound would find empty queues; not launching them saves their fixed cost.
static uint32_t segment_bound(const gpuart_hip_ctx *c, const gpuart_params *p) {
    if (p->maxSegments <= 0 || !(1.0f > p->minWeight)) return 0;
    uint32_t n = (uint32_t)p->maxSegments;
    if (!(p->minWeight > 0)) return n;
    static const float ALBEDO[4][3] = {{0.65f, 0.4f, 0.35f}, {0.1f, 0.2f, 0.1f}, {0.3f, 0.3f, 0.3f}, {0.3f, 0.3f, 0.3f}};
    // The user sphere shades as a sphere, and even with radius 0 (= "disabled") its quadratic can report a hit through
    // rounding (sphere.glsl:47-52), so the sphere albedo always takes part in the bound.
    const uint32_t types = c->type_mask | 1u;
    uint32_t bound = n;
    for (int ch = 0; ch < 3; ch++) {
        float a = 0;
        for (int t = 0; t < 4; t++) if (types & (1u << t)) a = std::max(a, ALBEDO[t][ch]);
        float w = 1.0f;
        uint32_t k = 0;
        while (k < n && w > p->minWeight * 1.0001f) { w *= a; k++; }
        bound = std::min(bound, k);
    }
    return std::max<uint32_t>(bound, 1);
}

/// Launches the collected passes as one run of the wavefront pipeline on the next pass lane.
namespace {
/// One run of the wavefront pipeline for the pending passes [first, first + count) on the next pass lane.
int launch_run(gpuart_hip_ctx *c, size_t first, size_t count) {
    int r;
    const gpuart_params *p = &c->pend_params;
    const int npaths = c->pend_npaths;
    Scene sc = scene_of(c);
    TimedLaunch t;
    PassLane &l = c->lanes[c->next_lane];
    c->next_lane = (c->next_lane + 1) % c->lanes_in_use;
    const uint32_t nseg = segment_bound(c, p);
    l.pb.batch = (uint32_t)count;
    SeedBatch seeds{};
    for (size_t k = 0; k < count; k++) seeds.seed[k] = c->pend_seeds[first + k];
    if ((r = ensure_segment_counters(c, l, nseg))) return r;
    const PathBuffers &b = l.pb;
    const bool refwork = c->mode == 1;
    const bool flat_only = c->lean_kernels && (c->type_mask & ~(uint32_t)GD_FLAT_TYPES) == 0;  // triangle meshes + discs
    const bool detail = c->timing_level >= 2;
    const dim3 pgrid(c->grid_waves);
    const dim3 sgrid(std::min<uint32_t>(c->grid_waves, b.n_slots * b.batch / BLOCK));
    int j_cur = 0;
    if (l.used) HIP_TRY(hipStreamWaitEvent(l.main, l.ev_free, 0));  // the lane's previous pass has been accumulated
    if ((r = begin_timed(c, t, 0, l.main))) return r;
    // one BVH-query launch: closest-hit queries of segment seg_c and / or Sun-shadow queries of segment seg_s
    auto trace = [&](int seg_c, int seg_s) -> int {
        TimedLaunch tt;
        int rr;
        if (detail && (rr = begin_timed(c, tt, 1, l.main))) return rr;
        if (refwork) k_trace<true, GD_ALL_TYPES><<<pgrid, BLOCK, 0, l.main>>>(sc, c->frame, *p, b, seg_c, seg_s, 0, j_cur, npaths, l.passcolor, l.spill_main, c->d_counters, c->tune);
        else if (flat_only) k_trace<false, GD_FLAT_TYPES><<<pgrid, BLOCK, 0, l.main>>>(sc, c->frame, *p, b, seg_c, seg_s, 1, j_cur, npaths, l.passcolor, l.spill_main, c->d_counters, c->tune);
        else k_trace<false, GD_ALL_TYPES><<<pgrid, BLOCK, 0, l.main>>>(sc, c->frame, *p, b, seg_c, seg_s, 1, j_cur, npaths, l.passcolor, l.spill_main, c->d_counters, c->tune);
        if (detail && (rr = end_timed(c, tt, l.main))) return rr;
        return 0;
    };
    for (int j = 0; j < npaths; j++) {
        j_cur = j;
        HIP_TRY(hipMemsetAsync(b.counters, 0, 4 * ((size_t)nseg + 1) * sizeof(uint32_t), l.main));
        k_gen<<<sgrid, BLOCK, 0, l.main>>>(c->frame, *p, seeds, j, npaths, b, l.passcolor);
        if (nseg && (r = trace(0, -1))) return r;
        for (uint32_t seg = 0; seg < nseg; seg++) {
            if (refwork) k_shade<true><<<sgrid, BLOCK, 0, l.main>>>(sc, c->frame, *p, seeds, b, (int)seg, j, npaths, l.passcolor, c->d_counters);
            else k_shade<false><<<sgrid, BLOCK, 0, l.main>>>(sc, c->frame, *p, seeds, b, (int)seg, j, npaths, l.passcolor, c->d_counters);
            // the Sun-shadow queries of this segment travel with the closest-hit queries of the next one
            const int next_c = seg + 1 < nseg ? (int)seg + 1 : -1, sh = p->sunEnabled == 1 ? (int)seg : -1;
            if ((next_c >= 0 || sh >= 0) && (r = trace(next_c, sh))) return r;
        }
        HIP_TRY(hipGetLastError());
    }
    if ((r = end_timed(c, t, l.main))) return r;
    // accumulate in pass order on the primary stream, then release the lane
    HIP_TRY(hipEventRecord(l.ev_done, l.main));
    HIP_TRY(hipStreamWaitEvent(c->stream, l.ev_done, 0));
    k_accumulate<<<dim3((unsigned)((c->tile_pixels + 255) / 256)), 256, 0, c->stream>>>(c->d_accum, l.passcolor, c->tile_pixels, b.batch);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(l.ev_free, c->stream));
    l.used = true;
    return 0;
}
}  // namespace

int gpuart_hip_flush(gpuart_hip_ctx *c) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    const size_t pending = c->pend_seeds.size();
    if (!pending) return 0;
    HIP_TRY(hipSetDevice(c->device));
    // What is still pending when something observes or changes state (read-back, finish, ...) is split into runs of
    // at least ~2M paths on separate pass lanes, so that the end of a pass sequence still overlaps its kernels.
    const size_t runs = std::max<size_t>(1, std::min<size_t>({(size_t)c->lanes_in_use, pending, pending * c->n_slots / c->min_run_paths}));
    int r = 0;
    for (size_t k = 0, first = 0; k < runs && !r; k++) {
        const size_t count = (pending - first) / (runs - k);
        r = launch_run(c, first, count);
        first += count;
    }
    c->pend_seeds.clear();
    return r;
}


int gpuart_hip_pt_pass(gpuart_hip_ctx *c, const gpuart_params *p, const float randSeed[4], int npaths) {
    int r = check_ready(c, p);
    if (r) return r;
    if (!randSeed || npaths < 0) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    if (npaths == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    const float4 seed = make_float4(randSeed[0], randSeed[1], randSeed[2], randSeed[3]);
    if (c->mode == 2) {  // megakernel: the whole path in one thread, on the primary stream (ablation / cross-check)
        if ((r = gpuart_hip_flush(c))) return r;
        Scene sc = scene_of(c);
        TimedLaunch t;
        if ((r = begin_timed(c, t, 0))) return r;
        dim3 grid(std::min<uint32_t>(c->grid_waves, c->n_slots / BLOCK));
        k_pt_mega<false><<<grid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, seed, npaths, c->n_slots, c->d_accum, c->d_spill, c->d_counters);
        HIP_TRY(hipGetLastError());
        return end_timed(c, t);
    }
    // Passes are collected and launched run_passes at a time (plan_runs). A pass with different parameters
    // starts a new batch; anything that observes or changes state flushes first.
    if (!c->pend_seeds.empty() && (memcmp(&c->pend_params, p, sizeof *p) != 0 || c->pend_npaths != npaths))
        if ((r = gpuart_hip_flush(c))) return r;
    c->pend_params = *p;
    c->pend_npaths = npaths;
    c->pend_seeds.push_back(seed);
    if (c->pend_seeds.size() >= c->run_passes) {
        r = launch_run(c, 0, c->pend_seeds.size());
        c->pend_seeds.clear();
        return r;
    }
    return 0;
}

int gpuart_hip_export(gpuart_hip_ctx *c, int which, void *rgba_device, float divide_by) {
    if (!c || !rgba_device || (which != 0 && which != 1) || !c->tile_pixels) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    const float4 *src = which == 0 ? c->d_direct : c->d_accum;
    if (!(divide_by > 0)) divide_by = 1.0f;
    size_t n = c->tile_pixels;
    k_scale_copy<<<dim3((unsigned)((n + 255) / 256)), 256, 0, c->stream>>>(src, (float4 *)rgba_device, n, divide_by);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gpuart_hip_read(gpuart_hip_ctx *c, int which, float *rgba_host, float divide_by) {
    if (!c || !rgba_host || (which != 0 && which != 1) || !c->tile_pixels) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    size_t bytes = c->tile_pixels * sizeof(float4);
    const float4 *src = which == 0 ? c->d_direct : c->d_accum;
    if (divide_by > 0 && divide_by != 1.0f) {
        int r = ensure_scratch(c, bytes);
        if (r) return r;
        r = gpuart_hip_export(c, which, c->d_scratch, divide_by);
        if (r) return r;
        src = (const float4 *)c->d_scratch;
    }
    HIP_TRY(hipMemcpyAsync(rgba_host, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int gpuart_hip_write(gpuart_hip_ctx *c, int which, const float *rgba_host) {
    if (!c || !rgba_host || which != 1 || !c->tile_pixels) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    HIP_TRY(hipMemcpyAsync(c->d_accum, rgba_host, c->tile_pixels * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int gpuart_hip_finish(gpuart_hip_ctx *c) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    return drain(c);
}

int gpuart_hip_set_mode(gpuart_hip_ctx *c, int mode) {
    if (!c || mode < 0 || mode > 2) return fail(GPUART_HIP_ERR_ARG, "bad mode");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    int r = drain(c);  // modes use different streams; keep their passes ordered
    if (r) return r;
    c->mode = mode;
    return 0;
}

int gpuart_hip_set_timing(gpuart_hip_ctx *c, int level) {
    if (!c || level < 0 || level > 2) return fail(GPUART_HIP_ERR_ARG, "bad timing level");
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    c->timing_level = level;
    return 0;
}

int gpuart_hip_counters(gpuart_hip_ctx *c, gpuart_counters *out, int reset) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    unsigned long long h[8];
    int rr = drain(c);
    if (rr) return rr;
    HIP_TRY(hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (out) {
        out->rays = h[0]; out->nodes = h[1];
        for (int k = 0; k < 4; k++) out->prim_tests[k] = h[2 + k];
        out->segments = h[6];
    }
    if (reset) HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof h, c->stream));
    return 0;
}

int gpuart_hip_kernel_time(gpuart_hip_ctx *c, int cls, double *total_ms, uint64_t *launches, int reset) {
    if (!c || cls < 0 || cls > 1) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    int r = fold_timings(c);
    if (r) return r;
    if (total_ms) *total_ms = c->timed_ms[cls];
    if (launches) *launches = c->timed_launches[cls];
    if (reset) { c->timed_ms[cls] = 0; c->timed_launches[cls] = 0; }
    return 0;
}

int gpuart_hip_scene_info(gpuart_hip_ctx *c, uint64_t *nodes, uint64_t *prims, uint32_t *max_depth, uint64_t *device_bytes) {
    if (!c || !c->have_scene) return fail(GPUART_HIP_ERR_ARG, "no scene uploaded");
    if (nodes) *nodes = c->n_nodes;
    if (prims) *prims = c->n_prims;
    if (max_depth) *max_depth = c->max_depth;
    if (device_bytes) *device_bytes = c->scene_bytes;
    return 0;
}

// ---- test hooks ------------------------------------------------------------------------------------
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), 256, 0, c->stream

int gpuart_hip_test_random(gpuart_hip_ctx *c, const float *in, int n, float *out) {
    const float *ins[] = {in}; float *outs[] = {out};
    return run_hook(c, n, ins, 1, outs, 1, [&](auto &i, auto &o) { k_test_random<<<GRID1(n)>>>(i[0], n, o[0]); });
}
int gpuart_hip_test_math(gpuart_hip_ctx *c, const float *in, int n, float *out) {
    const float *ins[] = {in}; float *outs[] = {out};
    return run_hook(c, n, ins, 1, outs, 1, [&](auto &i, auto &o) { k_test_math<<<GRID1(n)>>>(i[0], n, o[0]); });
}
int gpuart_hip_test_hemisphere(gpuart_hip_ctx *c, const float *v, const float *ri, int n, float *out) {
    const float *ins[] = {v, ri}; float *outs[] = {out};
    return run_hook(c, n, ins, 2, outs, 1, [&](auto &i, auto &o) { k_test_hemisphere<<<GRID1(n)>>>(i[0], i[1], n, o[0]); });
}
int gpuart_hip_test_inside_cone(gpuart_hip_ctx *c, const float *v, const float *nrm, const float *ri, float ha, int n, float *out) {
    const float *ins[] = {v, nrm, ri}; float *outs[] = {out};
    return run_hook(c, n, ins, 3, outs, 1, [&](auto &i, auto &o) { k_test_inside_cone<<<GRID1(n)>>>(i[0], i[1], i[2], ha, n, o[0]); });
}
int gpuart_hip_test_sky(gpuart_hip_ctx *c, const float *dir, const float sda[4], int n, float *out) {
    if (!sda) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    Float4Arg a; memcpy(a.v, sda, 16);
    const float *ins[] = {dir}; float *outs[] = {out};
    return run_hook(c, n, ins, 1, outs, 1, [&](auto &i, auto &o) { k_test_sky<<<GRID1(n)>>>(i[0], a, n, o[0]); });
}
int gpuart_hip_test_intersect(gpuart_hip_ctx *c, int ptype, const float *rs, const float *rd, const float *quads, int n,
                              float *out0, float *out1) {
    if (ptype < 0 || ptype > 3 || !quads || n < 0) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    // canonical payload (4 quads per sample) -> device records, through the uploader's packer
    std::vector<float4> recs((size_t)n * 3);
    for (int i = 0; i < n; i++) Converter::pack_prim((uint32_t)ptype, quads + 16 * (size_t)i, &recs[3 * (size_t)i]);
    // three record quads are passed as three n x 4 input arrays (de-interleaved), then re-read as records
    std::vector<float> packed((size_t)n * 12);
    memcpy(packed.data(), recs.data(), packed.size() * 4);
    const float *ins[] = {rs, rd, packed.data(), packed.data() + (size_t)n * 4, packed.data() + (size_t)n * 8};
    float *outs[] = {out0, out1};
    // inputs 2,3,4 are contiguous in device scratch, so i[2] addresses all 3n quads
    return run_hook(c, n, ins, 5, outs, 2, [&](auto &i, auto &o) { k_test_intersect<<<GRID1(n)>>>(i[0], i[1], i[2], n, o[0], o[1]); });
}
int gpuart_hip_test_aabb(gpuart_hip_ctx *c, const float *rs, const float *rd, const float *bmin, const float *bmax, int n, float *out) {
    const float *ins[] = {rs, rd, bmin, bmax}; float *outs[] = {out};
    return run_hook(c, n, ins, 4, outs, 1, [&](auto &i, auto &o) { k_test_aabb<<<GRID1(n)>>>(i[0], i[1], i[2], i[3], n, o[0]); });
}
int gpuart_hip_test_traverse(gpuart_hip_ctx *c, const float *rs, const float *rd, const float us[4], int n, int any_hit,
                             float *out0, float *out1) {
    if (!c || !c->have_scene || !us) return fail(GPUART_HIP_ERR_ARG, "no scene uploaded");
    Float4Arg a; memcpy(a.v, us, 16);
    Scene sc = scene_of(c);
    const float *ins[] = {rs, rd}; float *outs[] = {out0, out1};
    return run_hook(c, n, ins, 2, outs, 2, [&](auto &i, auto &o) {
        dim3 grid(std::min<uint32_t>(c->grid_waves, (uint32_t)((n + BLOCK - 1) / BLOCK)));
        if (any_hit) k_test_traverse<true><<<grid, BLOCK, 0, c->stream>>>(sc, i[0], i[1], a, n, o[0], o[1], c->d_spill);
        else k_test_traverse<false><<<grid, BLOCK, 0, c->stream>>>(sc, i[0], i[1], a, n, o[0], o[1], c->d_spill);
    });
}
int gpuart_hip_test_cam_rays(gpuart_hip_ctx *c, float *rstart, float *rdir) {
    if (!c || !c->have_camera || !c->tile_pixels) return fail(GPUART_HIP_ERR_ARG, "camera / frame not set");
    int n = (int)c->tile_pixels;
    float *outs[] = {rstart, rdir};
    Frame f = c->frame;
    return run_hook(c, n, nullptr, 0, outs, 2, [&](auto &, auto &o) { k_test_cam_rays<<<GRID1(n)>>>(f, o[0], o[1]); });
}

}  // extern "C"
