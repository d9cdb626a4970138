// ubench.hip — micro-benchmarks that calibrate the ceilings the BVH-query kernel is priced against (gfx950 only):
//   valu      : wave64 VALU issue rate per SIMD at 1..8 resident waves (independent / dependent v_fma_f32,
//               and the v_cmp -> v_cndmask / v_med3 mix of the AABB test)
//   aabb      : gd::aabb_entry (the product's own box test) on register-resident data: box tests per second with no memory
//   gather    : 64-byte node records fetched at data-dependent addresses (4 x global_load_dwordx4 per lane), the access
//               pattern of trav_step_box, over arrays of several sizes (L1 / L2 / Infinity-Cache resident)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../gpuart_amd/csrc/hip -I../../include -o ubench ubench.hip
// Prints one line per measurement; `ubench json` prints one JSON object (tools/ubench/run.sh stores it under profiles/).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "device_scene.h"

// Address chains must DEPEND on the loaded data (a traversal step's next address does) without being STEERED by it: with
// idx' = f(idx, data[idx & mask]) the low bits form one map r -> r' shared by every lane, all lanes fall into its few short cycles
// within a few hundred steps and the "random" gather turns into a few hundred hot records (round 2's first calibration had
// that flaw). g_zero is 0 at run time, unknown at compile time: `+ (h & zero)` keeps the dependency and leaves the sequence a
// full-period LCG per lane.
__device__ uint32_t g_zero = 0;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static int g_cus = 256;
static double g_clock_hz = 2.4e9;

// ---- VALU issue ---------------------------------------------------------------------------------
// 16 independent accumulators, ITER x 16 v_fma_f32 per lane
__global__ void __launch_bounds__(64) k_valu_indep(float *out, int iters, unsigned long long *cycles) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    float b = 1.0000001f, c = 1e-9f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
// the same with 128 v_fma_f32 (or v_add_f32 / v_mul_f32 in their 4-byte VOP2 encoding) between two loop branches: is the 16-instruction
// loop's rate the VALU's or the loop's?
template <int OP>
__global__ void __launch_bounds__(64) k_valu_long(float *out, int iters, unsigned long long *cycles) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    float b = 1.0000001f, c = 1e-9f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
__global__ void __launch_bounds__(64) k_valu_dep(float *out, int iters, unsigned long long *cycles) {
    float a = threadIdx.x * 0.001f;
    float b = 1.0000001f, c = 1e-9f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
// the select mix of the box test: k>=0 ? k : inf ; med3(v,lo,hi)==v ? c : inf  (4 independent chains, 16 VALU per round)
__global__ void __launch_bounds__(64) k_valu_select(float *out, int iters, unsigned long long *cycles) {
    float k[4], v[4];
    for (int i = 0; i < 4; i++) { k[i] = threadIdx.x * 0.01f - 0.3f + i; v[i] = 0.5f + 0.001f * threadIdx.x; }
    const float lo = 0.25f, hi = 0.75f, INF = __builtin_inff();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float c = (k[i] >= 0) ? k[i] : INF;                                   // v_cmp + v_cndmask
            c = (__builtin_amdgcn_fmed3f(v[i], lo, hi) == v[i]) ? c : INF;        // v_med3 + v_cmp + v_cndmask
            asm volatile("" : "+v"(c));
            k[i] = c * 0.999f;                                                    // v_mul
            v[i] = v[i] + 1e-7f;                                                  // v_add
            asm volatile("" : "+v"(k[i]), "+v"(v[i]));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = k[0] + k[1] + k[2] + k[3] + v[0] + v[1] + v[2] + v[3];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- the product's box test on registers ---------------------------------------------------------
template <bool QUICK = false>  // QUICK: the quick answer of box_quick.h (what the product runs for all but a few boxes in 100 000) instead of the six face tests
__global__ void __launch_bounds__(64) k_aabb(float *out, int iters, unsigned long long *cycles) {
    using namespace gd;
    Ray r;
    r.o = f3(0.1f + threadIdx.x * 0.01f, -3.0f, 1.0f);
    r.d = f3(0.02f * threadIdx.x - 0.6f, 1.0f, -0.1f);
    F3 rdiv = f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
    F3 bmin = f3(-1, -1, 0), bmax = f3(1, 1, 2);
    float acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        float e;
        bool h;
        if (QUICK) {
            const float cs = gq_ray_slack(gq_slack_of_tree(2.0f), r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, rdiv.x, rdiv.y, rdiv.z);  // (per step here: the ray is opaque per iteration)
            const bool sure = box_quick(r, rdiv, bmin, bmax, cs, e, h);
            acc += sure ? 0.0f : 3.0f;
        } else
            h = aabb_entry(r, rdiv, bmin, bmax, e);
        acc += h ? e : 1.0f;
        // keep the compiler from hoisting or simplifying any part of the test: every operand is opaque per iteration
        asm volatile("" : "+v"(bmin.x), "+v"(bmin.y), "+v"(bmin.z), "+v"(bmax.x), "+v"(bmax.y), "+v"(bmax.z), "+v"(acc));
        asm volatile("" : "+v"(r.o.x), "+v"(r.o.y), "+v"(r.o.z), "+v"(r.d.x), "+v"(r.d.y), "+v"(r.d.z));
        asm volatile("" : "+v"(rdiv.x), "+v"(rdiv.y), "+v"(rdiv.z));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- gather of 64-byte records -------------------------------------------------------------------
// chain: the next record index depends on the loaded data (as a traversal step's does); CHAINS independent chains per lane
template <int CHAINS>
__global__ void __launch_bounds__(64) k_gather(const float4 *__restrict__ recs, uint32_t mask, int iters, float *out,
                                               unsigned long long *cycles) {
    uint32_t idx[CHAINS];
    float acc = 0;
    const uint32_t zero = g_zero;
    for (int c = 0; c < CHAINS; c++) idx[c] = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + c * 40503u;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            const float4 *rec = recs + 4 * (size_t)(idx[c] & mask);
            float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
            asm volatile("" : "+v"(q0.w), "+v"(q1.w));
            acc += (q0.x + q1.y) + (q2.z + q3.x);
            idx[c] = idx[c] * 1664525u + 1013904223u + ((__float_as_uint(q0.w) + __float_as_uint(q1.w)) & zero);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc + idx[0];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- how the vector-memory pipeline prices one wave-wide dwordx4 load, by address pattern ------------------------------
// PATTERN 0: every lane its own random 64-B record, piece k in instruction k (what trav_step_box does: 64 distinct lines
//            per instruction);  1: quad-cooperative — in instruction k the four lanes of a quad read the four 16-B pieces
//            of the record wanted by the quad's k-th lane (16 distinct 64-B segments per instruction, same bytes in total);
//            2: all 64 lanes consecutive 16-B pieces (1 KB contiguous);  3: all lanes the same 16 bytes;
//            4: as 0 but only every 4th lane active (16 active lanes, 16 distinct lines)
template <int PATTERN>
__global__ void __launch_bounds__(64) k_loadcost(const float4 *__restrict__ recs, uint32_t mask, int iters, float *out,
                                                 unsigned long long *cycles) {
    const uint32_t lane = threadIdx.x, zero = g_zero;
    uint32_t idx = (blockIdx.x * 64 + lane) * 2654435761u;
    float acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (PATTERN == 5) {  // as 1 without the cross-lane moves: every lane follows the index streams of the four owners it loads for
        uint32_t ix[4];
        for (int k = 0; k < 4; k++) ix[k] = (blockIdx.x * 64 + 16 * k + (lane >> 2)) * 2654435761u;
        for (int it = 0; it < iters; it++) {
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = recs[4 * (size_t)(ix[k] & mask) + (lane & 3)];
            asm volatile("" : "+v"(q[0].w), "+v"(q[1].w), "+v"(q[2].w), "+v"(q[3].w));
            for (int k = 0; k < 4; k++) { asm volatile("" : "+v"(q[k].x), "+v"(q[k].y), "+v"(q[k].z), "+v"(q[k].w)); acc += (q[k].x + q[k].y) + q[k].z; }
            const uint32_t h = (__float_as_uint(q[0].w) + __float_as_uint(q[1].w) + __float_as_uint(q[2].w) + __float_as_uint(q[3].w)) & zero;
#pragma unroll
            for (int k = 0; k < 4; k++) ix[k] = ix[k] * 1664525u + 1013904223u + h;
        }
        idx = ix[0];
    } else
    if (PATTERN != 4 || (lane & 3) == 0)
    for (int it = 0; it < iters; it++) {
        float4 q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            size_t a;
            if (PATTERN == 0 || PATTERN == 4) a = 4 * (size_t)(idx & mask) + k;
            else if (PATTERN == 1) a = 4 * (size_t)((uint32_t)__shfl((int)idx, (int)((lane & ~3u) | k), 64) & mask) + (lane & 3);
            else if (PATTERN == 2) a = 4 * (size_t)(((idx & mask) & ~63u)) + 64 * k + lane;  // wave-uniform base would be ideal; close enough
            else a = 4 * (size_t)((uint32_t)__shfl((int)idx, 0, 64) & mask) + k;
            q[k] = recs[a];
        }
        asm volatile("" : "+v"(q[0].w), "+v"(q[1].w), "+v"(q[2].w), "+v"(q[3].w));
        for (int k = 0; k < 4; k++) { asm volatile("" : "+v"(q[k].x), "+v"(q[k].y), "+v"(q[k].z), "+v"(q[k].w)); acc += (q[k].x + q[k].y) + q[k].z; }
        idx = idx * 1664525u + 1013904223u + ((__float_as_uint(q[0].w) + __float_as_uint(q[1].w) + __float_as_uint(q[2].w) + __float_as_uint(q[3].w)) & zero);
        if (PATTERN == 2) idx = (uint32_t)__shfl((int)idx, 0, 64);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc + idx;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- what a divergent per-lane record fetch costs by instruction width: 64 B as 4 x dwordx4, 56 B as 3 x dwordx4 + dwordx2,
//      32 B as 2 x dwordx4, 64 B as 8 x dwordx2, 64 B as 16 x dword (every lane its own random 64-B record, dependent chain)
template <int SHAPE>
__global__ void __launch_bounds__(64) k_width(const float *__restrict__ recs, uint32_t mask, int iters, float *out, unsigned long long *cycles) {
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    const uint32_t zero = g_zero;
    float acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        const float *r = recs + 16 * (size_t)(idx & mask);
        float s = 0; uint32_t h = 0;
        if (SHAPE == 0) {
            float4 a = ((const float4 *)r)[0], b = ((const float4 *)r)[1], c = ((const float4 *)r)[2], d = ((const float4 *)r)[3];
            asm volatile("" : "+v"(a.w), "+v"(b.w), "+v"(c.w), "+v"(d.w));
            s = (a.x + b.y) + (c.z + d.x); h = __float_as_uint(a.w) + __float_as_uint(b.w) + __float_as_uint(c.w) + __float_as_uint(d.w);
        } else if (SHAPE == 1) {
            float4 a = ((const float4 *)r)[0], b = ((const float4 *)r)[1], c = ((const float4 *)r)[2]; float2 d = ((const float2 *)r)[6];
            asm volatile("" : "+v"(a.w), "+v"(b.w), "+v"(c.w), "+v"(d.y));
            s = (a.x + b.y) + (c.z + d.x); h = __float_as_uint(a.w) + __float_as_uint(b.w) + __float_as_uint(c.w) + __float_as_uint(d.y);
        } else if (SHAPE == 2) {
            float4 a = ((const float4 *)r)[0], b = ((const float4 *)r)[1];
            asm volatile("" : "+v"(a.w), "+v"(b.w));
            s = a.x + b.y; h = __float_as_uint(a.w) + __float_as_uint(b.w);
        } else if (SHAPE == 3) {
#pragma unroll
            for (int k = 0; k < 8; k++) { float2 a = ((const float2 *)r)[k]; asm volatile("" : "+v"(a.y)); s += a.x; h += __float_as_uint(a.y); }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) { float a = r[k]; asm volatile("" : "+v"(a)); s += a; h += __float_as_uint(a); }
        }
        acc += s;
        idx = idx * 1664525u + 1013904223u + (h & zero);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc + idx;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- the node fetch as shipped (2 x dwordx4 + 2 x dwordx3 of one 64-byte record) against 3 x dwordx4 of a 48-byte record
//      + 1 x dwordx2 from a separate array of child references
template <int SHAPE>
__global__ void __launch_bounds__(64) k_node(const float4 *__restrict__ recs, const uint2 *__restrict__ refs, uint32_t mask, int iters,
                                             float *out, unsigned long long *cycles) {
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    const uint32_t zero = g_zero;
    float acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        const uint32_t r = idx & mask;
        uint32_t h;
        if (SHAPE == 0) {
            const float4 *p = recs + 4 * (size_t)r;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            asm volatile("" : "+v"(a.w), "+v"(b.w));
            acc += ((a.x + a.y) + (a.z + b.x)) + ((b.y + b.z) + (c.x + c.y)) + ((c.z + d.x) + (d.y + d.z));
            h = __float_as_uint(a.w) + __float_as_uint(b.w);
        } else if (SHAPE == 2) {  // 4 x dwordx3 + 1 x dwordx2, all from one packed 64-byte record
            struct P3 { float x, y, z; };
            const char *base = (const char *)(recs + 4 * (size_t)r);
            const P3 a = *(const P3 *)(base), b = *(const P3 *)(base + 12), c = *(const P3 *)(base + 24), d = *(const P3 *)(base + 36);
            const uint2 f = *(const uint2 *)(base + 48);
            acc += ((a.x + a.y) + (a.z + b.x)) + ((b.y + b.z) + (c.x + c.y)) + ((c.z + d.x) + (d.y + d.z));
            h = f.x + f.y;
        } else if (SHAPE == 3) {  // 3 x dwordx4 + 1 x dwordx2 from one 64-byte record, the narrow one kept narrow by its type
            const float4 *p = recs + 4 * (size_t)r;
            float4 a = p[0], b = p[1], c = p[2];
            const uint2 f = *(const uint2 *)((const char *)p + 48);
            asm volatile("" : "+v"(a.w), "+v"(b.w), "+v"(c.w));
            acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w));
            h = f.x + f.y;
        } else if (SHAPE == 4 || SHAPE == 5) {  // 3 x dwordx4 and nothing else: a 48-byte record packed at stride 48 (4) or padded to 64 (5)
            const float4 *p = recs + (SHAPE == 4 ? 3 : 4) * (size_t)r;
            float4 a = p[0], b = p[1], c = p[2];
            asm volatile("" : "+v"(a.w), "+v"(b.w), "+v"(c.w));
            acc += ((a.x + a.y) + (a.z + b.x)) + ((b.y + b.z) + (c.x + c.y)) + (c.z + c.w);
            h = __float_as_uint(a.w) + __float_as_uint(b.w);
        } else if (SHAPE == 6) {  // 4 x dwordx4 (all 64 bytes), the next address depending on the first two only (as SHAPE 0)
            const float4 *p = recs + 4 * (size_t)r;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            asm volatile("" : "+v"(a.w), "+v"(b.w), "+v"(c.w), "+v"(d.w));
            acc += ((a.x + a.y) + (a.z + b.x)) + ((b.y + b.z) + (c.x + c.y)) + ((c.z + d.x) + (d.y + d.z)) + (c.w + d.w);
            h = __float_as_uint(a.w) + __float_as_uint(b.w);
        } else if (SHAPE == 8 || SHAPE == 9) {  // as SHAPE 0 (8) / SHAPE 6 (9), but record r keeps its k-th 16-byte piece in slot (k + r) & 3: one instruction's lanes spread over the 16-byte slots
            const float4 *p = recs + 4 * (size_t)r;
            float4 a = p[r & 3], b = p[(r + 1) & 3], c = p[(r + 2) & 3], d = p[(r + 3) & 3];
            if (SHAPE == 8) asm volatile("" : "+v"(a.w), "+v"(b.w)); else asm volatile("" : "+v"(a.w), "+v"(b.w), "+v"(c.w), "+v"(d.w));
            acc += ((a.x + a.y) + (a.z + b.x)) + ((b.y + b.z) + (c.x + c.y)) + ((c.z + d.x) + (d.y + d.z));
            if (SHAPE == 9) acc += c.w + d.w;
            h = __float_as_uint(a.w) + __float_as_uint(b.w);
        } else if (SHAPE == 7) {  // 2 x dwordx4 of a 32-byte record
            const float4 *p = recs + 2 * (size_t)r;
            float4 a = p[0], b = p[1];
            asm volatile("" : "+v"(a.w), "+v"(b.w));
            acc += ((a.x + a.y) + (a.z + b.x)) + (b.y + b.z);
            h = __float_as_uint(a.w) + __float_as_uint(b.w);
        } else {
            const float4 *p = recs + 3 * (size_t)r;
            float4 a = p[0], b = p[1], c = p[2];
            uint2 f = refs[r];
            asm volatile("" : "+v"(a.w), "+v"(b.w), "+v"(c.w));
            acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w));
            h = f.x + f.y;
        }
        idx = idx * 1664525u + 1013904223u + (h & zero);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc + idx;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- a pointer-free (heap-numbered) node array: W distinct 48-byte records in use, spread over 2^depth slots — what does the address
//      translation of a sparse 26-GB array cost a gather that the caches see exactly as they see a dense one? (profiles/r04/step_sensitivity.txt)
__global__ void __launch_bounds__(64) k_sparse(const char *__restrict__ base, uint32_t mask, uint32_t nslots, int iters, float *out, unsigned long long *cycles) {
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    const uint32_t zero = g_zero;
    float acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        const uint32_t r = idx & mask;                                        // which of the W records
        const uint32_t slot = __umulhi(r * 2654435761u, nslots);              // where it lives (a fixed pseudo-random slot per record)
        const float4 *p = (const float4 *)(base + (size_t)slot * 48);
        float4 a = p[0], b = p[1], c = p[2];
        asm volatile("" : "+v"(a.w), "+v"(b.w), "+v"(c.w));
        acc += ((a.x + a.y) + (a.z + b.x)) + ((b.y + b.z) + (c.x + c.y)) + (c.z + c.w);
        idx = idx * 1664525u + 1013904223u + ((__float_as_uint(a.w) + __float_as_uint(b.w)) & zero);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc + idx;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- do the node fetch and the box arithmetic overlap? One traversal step as the product runs it: fetch a record (2 x dwordx4 +
//      2 x dwordx3, every lane its own L1-resident record), two aabb_entry tests on the fetched boxes. MODE 0: both, 1: fetch only
//      (the boxes are not tested), 2: tests only (boxes from registers). If the hardware overlaps the two across the waves of a SIMD,
//      MODE 0 costs max(1, 2); if it cannot, their sum.
template <int MODE, bool QUICK = false>
__global__ void __launch_bounds__(64) k_step(const float4 *__restrict__ recs, uint32_t mask, int iters, float *out, unsigned long long *cycles) {
    using namespace gd;
    __shared__ float4 s_stage[256];
    __shared__ uint32_t s_nodes[64];
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    const uint32_t zero = g_zero;
    Ray r;
    r.o = f3(0.1f + threadIdx.x * 0.01f, -3.0f, 1.0f);
    r.d = f3(0.02f * threadIdx.x - 0.6f, 1.0f, -0.1f);
    const F3 rdiv = f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
    const float quick_cs = gq_ray_slack(gq_slack_of_tree(2.0f), r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, rdiv.x, rdiv.y, rdiv.z);
    F3 lo0 = f3(-1, -1, 0), hi0 = f3(1, 1, 2), lo1 = f3(-2, -1, 0), hi1 = f3(0.5f, 1, 3);
    float acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        uint32_t h = 0;
        if (MODE != 2) {
            float4 a, b, c, d;
            if (MODE >= 3) {
                // cooperative fetch: the four lanes of a quad load the four 16-byte pieces of ONE record per instruction (16 records per
                // instruction, 64-byte lines whole) straight into LDS; every lane then reads its own record back. MODE 3: + box tests, 4: fetch only
                const uint32_t lane = threadIdx.x;
                s_nodes[(lane & 15) * 4 + (lane >> 4)] = idx & mask;
                const uint4 own = *(const uint4 *)&s_nodes[(lane >> 2) * 4];
                const uint32_t piece = (lane & 3) ^ ((lane >> 3) & 3);
                const uint32_t on[4] = {own.x, own.y, own.z, own.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(recs + 4 * (size_t)on[k] + piece),
                                                     (__attribute__((address_space(3))) void *)(s_stage + 64 * k), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint32_t sw = (lane >> 1) & 3;
                a = s_stage[4 * lane + (0 ^ sw)]; b = s_stage[4 * lane + (1 ^ sw)]; c = s_stage[4 * lane + (2 ^ sw)]; d = s_stage[4 * lane + (3 ^ sw)];
            } else {
                const float4 *p = recs + 4 * (size_t)(idx & mask);
                a = p[0]; b = p[1]; c = p[2]; d = p[3];
            }
            asm volatile("" : "+v"(a.w), "+v"(b.w));
            h = __float_as_uint(a.w) + __float_as_uint(b.w);
            if (MODE == 0 || MODE == 3) {  // boxes from the record (random bits scaled into a sane range: the arithmetic is what matters)
                lo0 = f3(a.x * 1e-9f, a.y * 1e-9f, a.z * 1e-9f); hi0 = f3(b.x * 1e-9f + 1, b.y * 1e-9f + 1, b.z * 1e-9f + 1);
                lo1 = f3(c.x * 1e-9f, c.y * 1e-9f, c.z * 1e-9f); hi1 = f3(d.x * 1e-9f + 1, d.y * 1e-9f + 1, d.z * 1e-9f + 1);
            } else {
                acc += ((a.x + a.y) + (a.z + b.x)) + ((b.y + b.z) + (c.x + c.y)) + ((c.z + d.x) + (d.y + d.z));
            }
        }
        if (MODE != 1 && MODE != 4) {
            float e0, e1;
            bool h0, h1;
            if (QUICK) {  // the step as the product runs it since round 4: two quick answers (the ray's slack is computed where the ray changes, i.e. outside)
                const bool s0 = box_quick(r, rdiv, lo0, hi0, quick_cs, e0, h0), s1 = box_quick(r, rdiv, lo1, hi1, quick_cs, e1, h1);
                acc += (s0 & s1) ? 0.0f : 3.0f;
            } else {
                h0 = aabb_entry(r, rdiv, lo0, hi0, e0); h1 = aabb_entry(r, rdiv, lo1, hi1, e1);
            }
            acc += (h0 ? e0 : 1.0f) + (h1 ? e1 : 2.0f);
            asm volatile("" : "+v"(lo0.x), "+v"(lo0.y), "+v"(lo0.z), "+v"(hi0.x), "+v"(hi0.y), "+v"(hi0.z), "+v"(acc));
            asm volatile("" : "+v"(lo1.x), "+v"(lo1.y), "+v"(lo1.z), "+v"(hi1.x), "+v"(hi1.y), "+v"(hi1.z));
            if (MODE == 0 || MODE == 3) h += __float_as_uint(acc);  // the next address waits for the tests, as a traversal step's does
        }
        idx = idx * 1664525u + 1013904223u + (h & zero);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc + idx;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}


// ---- round 5: fewer requests per node visit. A node box is the union of its two children's boxes (reference src/bvh.cpp:41-72), so for
//      each of the 6 axis-sides at least one child's plane IS the parent's: of a record's 12 child planes only 6 are new. k_step32 is the
//      traversal step on a 32-byte record {6 new planes, 2 refs with 3 selector bits each}: the lane carries the current node's six plane
//      PARAMETERS k = (plane - o) * rdiv (what the quick box test consumes), computes the six new ones (12 VALU instead of 24) and deals
//      them out with 12 selects. A child that is STACKED needs its own six parameters again when it is popped: POP_PCT per cent of the
//      lanes (data-dependent, divergent) first fetch a 32-byte box of their own from a second array and recompute the six parameters,
//      as a pop would. MODE 0: fetch + decode + two quick box tests; 1: fetch only.
//      SAME: the node's own box sits in the second half of its own 64-byte record instead (same line, same working set as the shipped layout):
//      a lane that descends fetches the first 32 bytes, a lane that popped all 64.
template <int MODE, int POP_PCT, bool SAME = false>
__global__ void __launch_bounds__(64) k_step32(const float4 *__restrict__ recs, const float4 *__restrict__ boxes, uint32_t mask, int iters, float *out, unsigned long long *cycles) {
    using namespace gd;
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    const uint32_t zero = g_zero;
    Ray r;
    r.o = f3(0.1f + threadIdx.x * 0.01f, -3.0f, 1.0f);
    r.d = f3(0.02f * threadIdx.x - 0.6f, 1.0f, -0.1f);
    const F3 rdiv = f3(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
    const float cs = gq_ray_slack(gq_slack_of_tree(2.0f), r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, rdiv.x, rdiv.y, rdiv.z);
    float kp[6] = {-1.0f, 2.0f, -0.5f, 1.5f, -0.25f, 3.0f};  // the current node's plane parameters (x lo, x hi, y lo, y hi, z lo, z hi)
    float acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        uint32_t h = 0;
        const bool popped = POP_PCT > 0 && ((idx >> 9) % 100u) < (uint32_t)POP_PCT;  // this lane's node came off the stack: its own box, fetched; six parameters
        const float4 *p = recs + (SAME ? 4 : 2) * (size_t)(idx & mask);
        float4 lo = make_float4(0, 0, 0, 0), hi = lo;
        if (popped) {  // (the requests first, all of them; the arithmetic on the own box after the record's requests are out)
            const float4 *bx = SAME ? p + 2 : boxes + 2 * (size_t)((idx >> 3) & mask);
            lo = bx[0]; hi = bx[1];
        }
        float4 a = p[0], b = p[1];
        asm volatile("" : "+v"(a.w), "+v"(b.w));
        if (popped) {
            asm volatile("" : "+v"(lo.w), "+v"(hi.w));
            kp[0] = (lo.x * 1e-9f - r.o.x) * rdiv.x; kp[1] = (hi.x * 1e-9f + 1 - r.o.x) * rdiv.x;
            kp[2] = (lo.y * 1e-9f - r.o.y) * rdiv.y; kp[3] = (hi.y * 1e-9f + 1 - r.o.y) * rdiv.y;
            kp[4] = (lo.z * 1e-9f - r.o.z) * rdiv.z; kp[5] = (hi.z * 1e-9f + 1 - r.o.z) * rdiv.z;
            h += __float_as_uint(lo.w) + __float_as_uint(hi.w);
        }
        const uint32_t ra = __float_as_uint(a.w), rb = __float_as_uint(b.w);
        h += ra + rb;
        if (MODE == 0) {
            // six new parameters (the planes are random bits scaled into a sane range: the arithmetic is what matters)
            const float kn[6] = {(a.x * 1e-9f - r.o.x) * rdiv.x, (a.y * 1e-9f + 1 - r.o.x) * rdiv.x, (a.z * 1e-9f - r.o.y) * rdiv.y,
                                 (b.x * 1e-9f + 1 - r.o.y) * rdiv.y, (b.y * 1e-9f - r.o.z) * rdiv.z, (b.z * 1e-9f + 1 - r.o.z) * rdiv.z};
            // selector bit j (bits 25..27 of the two refs): the LOWER child takes the new plane of axis-side j, the upper one the parent's; clear: the other way round
            const uint32_t sel = ((ra >> 25) & 7u) | (((rb >> 25) & 7u) << 3);
            float kl[6], kh[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const bool s = (sel >> j) & 1u;
                kl[j] = s ? kn[j] : kp[j];
                kh[j] = s ? kp[j] : kn[j];
            }
            float e0, e1;
            bool h0, h1;
            const bool s0 = gq_box(kl[0], kl[1], kl[2], kl[3], kl[4], kl[5], fabsf(rdiv.x), fabsf(rdiv.y), fabsf(rdiv.z), cs, e0, h0);
            const bool s1 = gq_box(kh[0], kh[1], kh[2], kh[3], kh[4], kh[5], fabsf(rdiv.x), fabsf(rdiv.y), fabsf(rdiv.z), cs, e1, h1);
            acc += (s0 & s1) ? 0.0f : 3.0f;
            acc += (h0 ? e0 : 1.0f) + (h1 ? e1 : 2.0f);
            // descend into the nearer child: ITS parameters become the carried ones (the product does exactly this)
            const bool near_lo = !(e1 < e0);
#pragma unroll
            for (int j = 0; j < 6; j++) kp[j] = near_lo ? kl[j] : kh[j];
            asm volatile("" : "+v"(kp[0]), "+v"(kp[1]), "+v"(kp[2]), "+v"(kp[3]), "+v"(kp[4]), "+v"(kp[5]), "+v"(acc));
            h += __float_as_uint(acc);  // the next address waits for the tests, as a traversal step's does
        } else {
            acc += ((a.x + a.y) + (a.z + b.x)) + (b.y + b.z);
        }
        idx = idx * 1664525u + 1013904223u + (h & zero);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc + idx + kp[0];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

struct Result { std::string name; double value; std::string unit; std::string note; };
static std::vector<Result> g_results;

template <class F>
static void timed(const char *name, int waves_per_simd, double work_per_wave, const char *unit, F launch, const char *note = "") {
    const int waves = g_cus * 4 * waves_per_simd;
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)waves * 64 * 4));
    CHECK(hipMalloc(&cyc, (size_t)waves * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(waves, out, cyc);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(waves, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(waves);
    CHECK(hipMemcpy(h.data(), cyc, (size_t)waves * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= waves;
    const double total = work_per_wave * waves;
    const double rate = total / (ms * 1e-3);
    // per-SIMD cycles per unit of work, from the wall time and the nominal clock; and from the in-kernel counter
    const double cyc_wall = g_clock_hz * (ms * 1e-3) / (work_per_wave * waves_per_simd);
    const double cyc_wave = mean / work_per_wave;  // s_memtime ticks per unit of work per wave
    printf("%-34s waves/SIMD %d  %8.3f ms  %10.4g %s/s   %6.2f clk/unit/SIMD (wall @%.2f GHz)   %7.2f memtime-ticks/unit/wave  %s\n",
           name, waves_per_simd, ms, rate, unit, cyc_wall, g_clock_hz / 1e9, cyc_wave, note);
    char key[128];
    snprintf(key, sizeof key, "%s.w%d", name, waves_per_simd);
    g_results.push_back({key, rate, std::string(unit) + "/s", note});
    snprintf(key, sizeof key, "%s.w%d.clk_per_unit_per_simd", name, waves_per_simd);
    g_results.push_back({key, cyc_wall, "cycles", "wall time x nominal clock"});
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    const bool json = argc > 1 && !strcmp(argv[1], "json");
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    g_clock_hz = prop.clockRate * 1e3;
    printf("device %s, %d CUs, clock %.0f MHz, wall-clock counter %d kHz\n", prop.gcnArchName, g_cus, prop.clockRate / 1e3, prop.clockRate);
    const int iters = 4096;
    {   // bring the clocks up before anything is timed (the first kernels of a process run at idle clocks)
        float *o; unsigned long long *cy;
        CHECK(hipMalloc(&o, (size_t)g_cus * 32 * 64 * 4)); CHECK(hipMalloc(&cy, (size_t)g_cus * 32 * 8));
        for (int r = 0; r < 40; r++) k_valu_indep<<<g_cus * 32, 64>>>(o, 8192, cy);
        CHECK(hipDeviceSynchronize());
        CHECK(hipFree(o)); CHECK(hipFree(cy));
    }
    const bool only_node = getenv("UBENCH_ONLY_NODE") != nullptr;
    if (!only_node) {
    for (int w : {1, 2, 4, 6, 8}) {
        timed("valu_fma_independent", w, iters * 16.0, "wave-instr", [&](int n, float *o, unsigned long long *c) { k_valu_indep<<<n, 64>>>(o, iters, c); });
    }
    for (int w : {2, 4, 6, 8}) {
        timed("valu_fma_independent_unrolled128", w, iters * 16.0, "wave-instr", [&](int n, float *o, unsigned long long *c) { k_valu_long<0><<<n, 64>>>(o, iters, c); });
        timed("valu_add_vop2_unrolled128", w, iters * 16.0, "wave-instr", [&](int n, float *o, unsigned long long *c) { k_valu_long<1><<<n, 64>>>(o, iters, c); });
        timed("valu_mul_vop2_unrolled128", w, iters * 16.0, "wave-instr", [&](int n, float *o, unsigned long long *c) { k_valu_long<2><<<n, 64>>>(o, iters, c); });
    }
    for (int w : {1, 2, 4, 6, 8}) {
        timed("valu_fma_dependent", w, iters * 16.0, "wave-instr", [&](int n, float *o, unsigned long long *c) { k_valu_dep<<<n, 64>>>(o, iters, c); });
    }
    for (int w : {1, 2, 4, 6, 8}) {
        timed("valu_select_mix", w, iters * 24.0, "wave-instr", [&](int n, float *o, unsigned long long *c) { k_valu_select<<<n, 64>>>(o, iters, c); },
              "4 chains x (cmp, med3, cmp, cndmask, mul, add): 24 VALU + 6 SALU per round");
    }
    for (int w : {1, 2, 4, 6, 8}) {
        timed("aabb_entry_registers", w, (double)iters, "box-tests(x64 lanes)", [&](int n, float *o, unsigned long long *c) { k_aabb<false><<<n, 64>>>(o, iters, c); });
        timed("box_quick_registers", w, (double)iters, "box-tests(x64 lanes)", [&](int n, float *o, unsigned long long *c) { k_aabb<true><<<n, 64>>>(o, iters, c); });
    }
    // gather: arrays of 2^k records of 64 B
    for (uint32_t log2n : {8u, 16u, 20u}) {  // 16 KB (L1), 512 KB (L2), 4 MB (one L2), 64 MB (Infinity Cache), 512 MB (HBM)
        const size_t nrec = (size_t)1 << log2n;
        float4 *recs;
        CHECK(hipMalloc(&recs, nrec * 64));
        std::vector<uint32_t> h(nrec * 16);
        uint32_t s = 12345;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = s >> 3; }
        CHECK(hipMemcpy(recs, h.data(), nrec * 64, hipMemcpyHostToDevice));
        char name[64];
        for (int w : {2, 6, 8}) {
            snprintf(name, sizeof name, "gather64B_dep1_%zuKB", nrec * 64 / 1024);
            timed(name, w, 512.0, "wave-gathers", [&](int n, float *o, unsigned long long *c) { k_gather<1><<<n, 64>>>(recs, (uint32_t)nrec - 1, 512, o, c); },
                  "one dependent chain per lane");
        }
        for (int w : {6}) {
            snprintf(name, sizeof name, "gather64B_dep4_%zuKB", nrec * 64 / 1024);
            timed(name, w, 512.0 * 4, "wave-gathers", [&](int n, float *o, unsigned long long *c) { k_gather<4><<<n, 64>>>(recs, (uint32_t)nrec - 1, 512, o, c); },
                  "four independent chains per lane");
        }
        CHECK(hipFree(recs));
    }
    }
    if (getenv("UBENCH_SPARSE")) {  // UBENCH_SPARSE=1: only this part
        struct Case { uint32_t log2w, log2slots; } cases[] = {{16, 16}, {16, 24}, {20, 20}, {20, 29}};
        for (const Case &cs : cases) {
            const size_t bytes = ((size_t)1 << cs.log2slots) * 48;
            char *base;
            if (hipMalloc(&base, bytes) != hipSuccess) { printf("sparse: no %zu MB\n", bytes >> 20); continue; }
            CHECK(hipMemset(base, 0, bytes));
            char name[96];
            snprintf(name, sizeof name, "sparse48B_3x4_%uK_records_in_%zuMB", (1u << cs.log2w) >> 10, bytes >> 20);
            for (int w : {2, 6})
                timed(name, w, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_sparse<<<n, 64>>>(base, (1u << cs.log2w) - 1, 1u << cs.log2slots, 1024, o, c); });
            CHECK(hipFree(base));
        }
        return 0;
    }
    for (uint32_t log2rec : {8u, 13u, 16u}) {  // 16 KB (vector L1), 512 KB, 4 MB (one XCD's L2)
        const size_t nrec = (size_t)1 << log2rec;
        char sz[32]; snprintf(sz, sizeof sz, "_%zuKB", nrec * 64 / 1024);
        auto nm = [&](const char *base) { static std::string keep; keep = std::string(base) + sz; return keep.c_str(); };
        float4 *recs;
        CHECK(hipMalloc(&recs, nrec * 64));
        std::vector<uint32_t> h(nrec * 16);
        uint32_t sd = 777;
        for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = sd >> 3; }
        CHECK(hipMemcpy(recs, h.data(), nrec * 64, hipMemcpyHostToDevice));
        const uint32_t m = (uint32_t)nrec - 1;
        float4 *boxes2;  // the pops' boxes of k_step32: a second array of the same size (same random contents)
        CHECK(hipMalloc(&boxes2, nrec * 64));
        CHECK(hipMemcpy(boxes2, h.data(), nrec * 64, hipMemcpyHostToDevice));
        for (int w : {2, 6}) {
            if (only_node ? w != 6 : log2rec != 16u) continue;
            timed(nm("load4x16B_lane_divergent"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_loadcost<0><<<n, 64>>>(recs, m, 1024, o, c); }, "64 lanes x own 64-B record");
            timed(nm("load4x16B_quad_cooperative"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_loadcost<1><<<n, 64>>>(recs, m, 1024, o, c); }, "same bytes, a quad reads one record per instruction");
            timed(nm("load4x16B_quad_cooperative_no_shuffles"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_loadcost<5><<<n, 64>>>(recs, m, 1024, o, c); }, "as above, addresses without cross-lane moves");
            timed(nm("load4x16B_contiguous"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_loadcost<2><<<n, 64>>>(recs, m, 1024, o, c); }, "1 KB contiguous per instruction");
            timed(nm("load4x16B_broadcast"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_loadcost<3><<<n, 64>>>(recs, m, 1024, o, c); }, "all lanes one record");
            timed(nm("load4x16B_16_lanes"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_loadcost<4><<<n, 64>>>(recs, m, 1024, o, c); }, "16 active lanes, own records");
        }
        for (int w : {6}) {
            if (log2rec != 16u) break;
            const float *rf = (const float *)recs;
            timed(nm("fetch64B_4x_dwordx4"), w, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_width<0><<<n, 64>>>(rf, m, 1024, o, c); });
            timed(nm("fetch56B_3x_dwordx4_1x_dwordx2"), w, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_width<1><<<n, 64>>>(rf, m, 1024, o, c); });
            timed(nm("fetch32B_2x_dwordx4"), w, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_width<2><<<n, 64>>>(rf, m, 1024, o, c); });
            timed(nm("fetch64B_8x_dwordx2"), w, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_width<3><<<n, 64>>>(rf, m, 1024, o, c); });
            timed(nm("fetch64B_16x_dword"), w, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_width<4><<<n, 64>>>(rf, m, 1024, o, c); });
        }
        {
            uint2 *refs;
            CHECK(hipMalloc(&refs, nrec * 8));
            CHECK(hipMemcpy(refs, h.data(), nrec * 8, hipMemcpyHostToDevice));
            for (int w : {2, 4, 8})
                timed(nm("node_2x4_2x3_one_record"), w, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<0><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_2x4_2x3_one_record"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<0><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_3x4_plus_refs_x2"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<1><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_4x3_1x2_one_record"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<2><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_3x4_1x2_one_record"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<3><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_3x4_48B_stride48"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<4><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_3x4_48B_stride64"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<5><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_4x4_64B"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<6><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_2x4_2x3_rotated_slots"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<8><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_2x4_2x3_rotated_slots"), 2, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<8><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_4x4_rotated_slots"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<9><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            timed(nm("node_2x4_32B"), 6, 1024.0, "wave-fetches", [&](int n, float *o, unsigned long long *c) { k_node<7><<<n, 64>>>(recs, refs, m, 1024, o, c); });
            CHECK(hipFree(refs));
        }
        {
            for (int w : {2, 4, 6, 8}) {
                if (log2rec != 8u && w != 6) continue;
                timed(nm("step_fetch_and_2_box_tests"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step<0><<<n, 64>>>(recs, m, 1024, o, c); });
                timed(nm("step_fetch_and_2_quick_box_tests"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step<0, true><<<n, 64>>>(recs, m, 1024, o, c); });
                timed(nm("step_fetch_only"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step<1><<<n, 64>>>(recs, m, 1024, o, c); });
                timed(nm("step_coop_fetch_and_2_box_tests"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step<3><<<n, 64>>>(recs, m, 1024, o, c); });
                timed(nm("step_coop_fetch_only"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step<4><<<n, 64>>>(recs, m, 1024, o, c); });
                timed(nm("step_coop_fetch_and_2_quick_box_tests"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step<3, true><<<n, 64>>>(recs, m, 1024, o, c); });
                // round 5: the 32-byte implicit-plane record (k_step32) over the same bytes: 2 x as many records, a second array of the same size for the pops' boxes
                timed(nm("step_fetch32B_only"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step32<1, 0><<<n, 64>>>(recs, recs, 2 * m + 1, 1024, o, c); });
                timed(nm("step_fetch32B_implicit_and_2_quick_box_tests_pop0"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step32<0, 0><<<n, 64>>>(recs, boxes2, 2 * m + 1, 1024, o, c); });
                timed(nm("step_fetch32B_implicit_and_2_quick_box_tests_pop25"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step32<0, 25><<<n, 64>>>(recs, boxes2, 2 * m + 1, 1024, o, c); });
                timed(nm("step_fetch32B_implicit_and_2_quick_box_tests_pop50"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step32<0, 50><<<n, 64>>>(recs, boxes2, 2 * m + 1, 1024, o, c); });
                // the same with the popped node's own box in the second half of its own 64-byte record (no second array, the shipped working set)
                timed(nm("step_fetch32of64B_implicit_and_2_quick_box_tests_pop0"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step32<0, 0, true><<<n, 64>>>(recs, recs, m, 1024, o, c); });
                timed(nm("step_fetch32of64B_implicit_and_2_quick_box_tests_pop25_same_line"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step32<0, 25, true><<<n, 64>>>(recs, recs, m, 1024, o, c); });
                timed(nm("step_fetch32of64B_implicit_and_2_quick_box_tests_pop40_same_line"), w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step32<0, 40, true><<<n, 64>>>(recs, recs, m, 1024, o, c); });
                timed("step_2_box_tests_only", w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step<2><<<n, 64>>>(recs, m, 1024, o, c); });
                timed("step_2_quick_box_tests_only", w, 1024.0, "wave-steps", [&](int n, float *o, unsigned long long *c) { k_step<2, true><<<n, 64>>>(recs, m, 1024, o, c); });
            }
        }
        CHECK(hipFree(recs));
        CHECK(hipFree(boxes2));
    }
    if (json) {
        printf("{");
        for (size_t i = 0; i < g_results.size(); i++)
            printf("%s\"%s\": %.6g", i ? ", " : "", g_results[i].name.c_str(), g_results[i].value);
        printf("}\n");
    }
    return 0;
}
