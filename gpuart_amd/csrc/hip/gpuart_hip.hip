// gpuart_hip.hip — context, scheduling and the C-ABI launcher of libgpuart_hip.so (gfx950 only); the kernels live in
// kernels_pipeline.h / kernels_test.h, the tree re-layout in converter.h.
// See include/gpuart_hip.h for the boundary and DESIGN.md for layout / kernel notes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "device_shade.h"
#include "gpuart_hip.h"
#include "gpuart_hip_test.h"

#include "kernels_pipeline.h"
#include "kernel_run.h"
#ifdef GPUART_HIP_TEST_HOOKS
#include "kernels_test.h"  // kernels of the device-function hooks (include/gpuart_hip_test.h)
#endif
#include "converter.h"
#include "gather.h"
#include "run_planner.h"
#include "bounded.h"

// =================================================================================================
// Host side: context, upload, launches
// =================================================================================================
namespace {

thread_local std::string g_last_error;

// Every pipeline run in flight has its own stream (+ the primary stream). ROCm maps HIP streams onto GPU_MAX_HW_QUEUES
// hardware queues (default 4) and kernels of streams that share a queue serialise, which would undo the overlap
// the pass lanes exist for; ask for more queues unless the user has chosen a value. Must happen before the HIP
// runtime initialises, hence a load-time constructor (bench.py also sets it before importing torch).
// Measured in round 5 (rocprofv3 kernel traces, profiles/r05/planner_even_runs.txt): whatever the value, the streams of a process share
// EIGHT hardware queues on this platform — the primary stream's and seven more, so the eighth pass lane shares one (a stream created
// with a priority gets a queue of its own: tried for the primary stream, nine queues, same times). Sharing by itself costs nothing
// measurable; sequences of seven runs that rotated over eight lanes were 1-3 % slower than on seven, so drain() restarts the rotation.
__attribute__((constructor)) void request_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "32", 0); }

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    bounded_ns::log_error(code, msg);  // (all threads' recent errors: what the phase watchdog prints, bounded.h)
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
    uint32_t *run_cursor = nullptr;  ///< chunk cursor of the lane's k_run launch
    hipEvent_t ev_done = nullptr;  ///< the run has finished (main stream)
    hipEvent_t ev_free = nullptr;  ///< its colour has been accumulated (primary stream): the lane may be reused
    bool used = false;
};

struct gpuart_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  ///< primary stream: accumulation (in pass order), direct lighting, copies, test hooks
    Frame frame{};
    bool have_camera = false, have_scene = false;
    bool empty_share = false;      ///< gpuart_hip_set_share with th == 0: nothing to render, still a member of the gather
    float4 *d_recs = nullptr, *d_prims = nullptr;
    uint32_t *d_cursor = nullptr;  ///< pixel cursor of k_direct_persistent
    uint4 *d_spill = nullptr;      ///< [spill_levels][grid_lanes] traversal-stack overflow for kernels on the primary stream
    // Birth order of the paths of a run (Frame::tile_order, k_tile_order): the tile's 8x8 blocks, most expensive first. The first run
    // after the tile was (re)allocated — and the first after every change of camera or scene — counts the shaded segments per block
    // (`gathers`), a sort behind it writes the buffer that is not in use, and the runs launched once that has finished use it.
    uint32_t *d_tile_order[2] = {nullptr, nullptr};
    uint32_t *d_tile_cost = nullptr;
    hipEvent_t ev_order = nullptr;     ///< the sort into d_tile_order[order_next] has finished
    int order_cur = -1, order_next = 0; ///< buffer in use (-1: none yet, row-major), buffer being written
    bool order_sorting = false;        ///< a gathering run + sort is under way
    bool order_stale = true;           ///< the order in use (or none) was not counted with this camera / scene
    bool order_auto = true;            ///< GPUART_HIP_TILE_ORDER (default 1)
    bool order_from_hook = false;      ///< gpuart_hip_test_tile_order set Frame::tile_order: it stays until the hook clears it
    std::vector<PassLane> lanes;
    uint32_t next_lane = 0;
    uint32_t spill_levels = 0;
    uint32_t num_cus = 256;
    uint32_t grid_waves = 4096;    ///< persistent grid: one wave per block
    uint32_t direct_waves = 4096;  ///< grid of k_direct_persistent (runs alone: needs the whole occupancy itself)
    uint32_t shade_waves = 4096;   ///< grid of the streaming kernels (k_gen, k_shade): latency-bound, so more waves than k_trace
    uint32_t run_waves = 5120;     ///< grid of k_run: fills every SIMD by itself (5 waves per SIMD)
    TraceTuning tune{128, 16, 16, 3};
    bool lean_kernels = true;      ///< use the BVH-query kernels specialised for the primitive types present
    // passes requested through gpuart_hip_pt_pass but not launched yet (same params, one seed each)
    std::vector<float4> pend_seeds;
    gpuart_params pend_params{};
    int pend_npaths = 0;
    uint64_t n_nodes = 0, n_prims = 0, scene_bytes = 0;
    uint32_t ref_order = 0;       ///< a small tree of regular boxes: the fast kernels keep the reference's order (GD_REF_ORDER variants)
    uint32_t nearest_min_prims = 0xffffffffu;  ///< GPUART_HIP_NEAREST_MIN_PRIMS / gpuart_hip_set_nearest_first: trees with at least this many primitives are
                                                ///< walked nearer child first (OPT-IN since round 5: that walk is not the reference's on phantom hits,
                                                ///< device_scene.h); default: never — every walk keeps the reference's order
    uint32_t type_mask = 0;  ///< bit t set: the scene holds primitives of type t
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    uint32_t root_ref = 0;
    bool chunk_from_env = false;  ///< GPUART_HIP_CHUNK was given: no per-launch choice of the chunk size
    float box_slack = __builtin_inff();  ///< box_quick.h's slack constant of the uploaded tree (+inf — also before any upload —: quick box answers are never taken)
    uint32_t quick_boxes = 1;     ///< GPUART_HIP_QUICK_BOXES (0: box_slack stays +inf — every box test runs its six face tests)
    uint32_t exact_boxes = 0;     ///< the uploaded tree holds an irregular box, or a box that does not bound what it holds (converter.h):
                                  ///< box tests take the comparison form and every walk keeps the reference's order
    uint32_t max_depth = 0;
    float4 *d_direct = nullptr, *d_accum = nullptr;
    unsigned long long *d_counters = nullptr;
    /// Tile size, run lengths, lanes in use, execution mode (gpuart_hip_set_mode: 0 fast — launch pipeline, or k_run for a small
    /// sequence —, 1 reference work + counters (k_run), 2 megakernel, 3 launch pipeline always, 4 fast with counters of the
    /// executed work (k_run), 5 k_run always) and the passes collected but not launched: run_planner.h
    RunPlanner plan;
    std::vector<TimedLaunch> pending, free_events;
    double timed_ms[2] = {0, 0};
    uint64_t timed_launches[2] = {0, 0};
    int timing_level = 1;  ///< 0 none, 1 per render call, 2 also per BVH-query kernel
    float *d_scratch = nullptr;  // test hooks
    size_t scratch_bytes = 0;
    // multi-GPU gather (gather.h)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_nranks = 0;
    bool comm_owned = false;          ///< made by gpuart_hip_comm_init / _init_all (destroyed with the context)
    float4 *d_send = nullptr, *d_stage = nullptr;
    size_t send_pixels = 0, stage_pixels = 0;
    void *d_hello = nullptr;          ///< [1 + nranks] GatherHello: own share + status, then everybody's (ncclAllGather)
    void *h_frame = nullptr;          ///< pinned landing area of gpuart_hip_gather_all_read's read-back (the bounded wait comes after the copy is queued)
    size_t h_frame_bytes = 0;
    void *h_hello = nullptr;          ///< pinned host copy of that table: the exchange's device-to-host copy must be truly asynchronous
                                      ///< (the bounded wait comes AFTER it is queued) and its target must outlive a timed-out call
    uint64_t comm_group = 0;          ///< which communicator this context is a rank of: contexts joined by one _comm_init_all (or one
                                      ///< unique id) share it; gpuart_hip_gather_all refuses ranks of different communicators
    uint32_t gather_timeout_ms = 60000;  ///< GPUART_HIP_GATHER_TIMEOUT_MS: bound on the host-side wait for the peers (0: none)
    bool abandoned = false;           ///< a bounded wait on the primary stream ran out: what is queued there may never complete, so
                                      ///< every later wait of this context (comm_destroy, destroy) is bounded too
};

namespace {

/// Waits for everything this context has enqueued, on all of its streams.
int drain(gpuart_hip_ctx *c) {
    for (auto &l : c->lanes) {
        if (l.main) HIP_TRY(hipStreamSynchronize(l.main));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->next_lane = 0;  // every lane is idle: the next sequence starts on the first one again (a plan of seven runs stays off the eighth lane, see above)
    return 0;
}


/// Waits for the context's primary stream, but not for ever: a peer that never arrives must not hang this process.
int wait_stream(gpuart_hip_ctx *c, uint32_t timeout_ms, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) { c->abandoned = false; return 0; }
        if (e != hipErrorNotReady) return fail(GPUART_HIP_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
        if (timeout_ms && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) {
            c->abandoned = true;  // what is queued on the stream may never complete: later waits of this context stay bounded (destroy included)
            return fail(GPUART_HIP_ERR_TIMEOUT, std::string(what) + ": not complete after " + std::to_string(timeout_ms) +
                        " ms (rank " + std::to_string(c->comm_rank) + " of " + std::to_string(c->comm_nranks) + "; a peer has not joined the collective)");
        }
        // (a gather's waits sit inside bench.py's timed region, where a rank's whole job is ~3 ms at N = 8: the first few milliseconds are
        //  polled without sleeping — a sleep of 50 us returns after 100-150 —, a wait that lasts longer is not in a hurry)
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else std::this_thread::yield();
    }
}
}  // namespace
extern "C" int gpuart_hip_flush(gpuart_hip_ctx *c);
namespace {

int realloc_tile(gpuart_hip_ctx *c) {
    int r = gpuart_hip_flush(c);
    if (r) return r;
    if ((r = drain(c))) return r;
    c->empty_share = false;
    for (auto &o : c->d_tile_order) if (o) { (void)hipFree(o); o = nullptr; }
    if (c->d_tile_cost) { (void)hipFree(c->d_tile_cost); c->d_tile_cost = nullptr; }
    c->frame.tile_order = nullptr; c->frame.tile_cost = nullptr;
    c->order_cur = -1; c->order_sorting = false; c->order_stale = true; c->order_from_hook = false;
    if (c->d_direct) { (void)hipFree(c->d_direct); c->d_direct = nullptr; }
    if (c->d_accum) { (void)hipFree(c->d_accum); c->d_accum = nullptr; }
    c->plan.tile_pixels = (size_t)c->frame.tw * c->frame.th;
    if (!c->plan.tile_pixels) { c->plan.set_tile(0, 0); return 0; }
    HIP_TRY(hipMalloc(&c->d_direct, c->plan.tile_pixels * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->d_accum, c->plan.tile_pixels * sizeof(float4)));
    HIP_TRY(hipMemsetAsync(c->d_direct, 0, c->plan.tile_pixels * sizeof(float4), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_accum, 0, c->plan.tile_pixels * sizeof(float4), c->stream));
    // wavefront path state: one slot per pixel of the 8x8-tile-padded tile, per pass in flight
    const size_t tiles = (size_t)((c->frame.tw + 7) / 8) * ((c->frame.th + 7) / 8);
    const size_t n = tiles * 64;
    if (n > 0xfffffff0ull) return fail(GPUART_HIP_ERR_ARG, "tile too large");
    // Several passes run through the pipeline together (slot = pass x pixel) while that stays within `batch_paths`
    // paths: a persistent k_trace wave then takes many rays per lane, and the drain at the end of every launch — waves
    // finishing their last, long rays with few lanes busy — shrinks relative to the useful work (SQ_INSTS_VALU per
    // ray falls by a quarter from 2M to 16M paths per launch at 1080p). run_planner.h decides how many.
    if (n * std::max<size_t>(1, std::min<size_t>(c->plan.batch_limit, c->plan.batch_paths / n)) > 0x7ffffff0ull)
        return fail(GPUART_HIP_ERR_ARG, "tile too large");  // two queues share one 32-bit index space in k_trace
    for (auto &l : c->lanes) {
        if (l.pathmem) { (void)hipFree(l.pathmem); l.pathmem = nullptr; }
        l.used = false;
    }
    c->next_lane = 0;
    c->pend_seeds.clear();
    c->plan.set_tile((uint32_t)n, c->plan.tile_pixels);
    // Lanes within the memory budget; when the device cannot give that much (other tenants), fewer lanes and then
    // smaller runs are tried before giving up — results do not depend on either.
    for (;;) {
        const size_t bytes = c->plan.lane_bytes(c->plan.max_batch);
        size_t got = 0;
        while (got < c->plan.lanes_in_use && hipMalloc(&c->lanes[got].pathmem, bytes) == hipSuccess) got++;
        if (got == c->plan.lanes_in_use) break;
        (void)hipGetLastError();  // clear the out-of-memory error
        for (size_t li = 0; li < got; li++) { (void)hipFree(c->lanes[li].pathmem); c->lanes[li].pathmem = nullptr; }
        if (!c->plan.shrink()) return fail(GPUART_HIP_ERR_DEVICE, "out of device memory for the path state of one pass");
    }
    const size_t B = c->plan.max_batch, lanes = c->plan.lanes_in_use;
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
        l.passcolor = (float4 *)m; m += B * c->plan.tile_pixels * sizeof(float4);
        b.hit = (uint2 *)m; m += nb * sizeof(uint2);
        b.queue[0] = (uint32_t *)m; m += nb * sizeof(uint32_t);
        b.queue[1] = (uint32_t *)m; m += nb * sizeof(uint32_t);
        b.shadow_queue = (uint32_t *)m;
        b.n_slots = (uint32_t)n;
        b.batch = 1;
        b.tile_pixels = (uint32_t)c->plan.tile_pixels;
    }
    return 0;
}

int ensure_segment_counters(gpuart_hip_ctx *c, PassLane &l, uint32_t nseg) {
    if (l.pb.counters && l.counter_segments >= nseg) return 0;
    if (l.pb.counters) { int r = drain(c); if (r) return r; (void)hipFree(l.pb.counters); l.pb.counters = nullptr; }
    HIP_TRY(hipMalloc(&l.pb.counters, 20 * ((size_t)nseg + 1) * sizeof(uint32_t)));  // 4 words per segment + 16 per-XCD cursors (experiment)
    l.pb.xcd_cursors = l.pb.counters + 4 * ((size_t)nseg + 1);
    l.counter_segments = nseg;
    return 0;
}

int ensure_spill(gpuart_hip_ctx *c) {
    const uint32_t levels = c->max_depth > GD_RING ? c->max_depth - GD_RING : 0;
    if (c->d_spill && c->spill_levels >= levels) return 0;
    int r = drain(c);
    if (r) return r;
    const size_t bytes = ((size_t)levels + 1) * std::max({c->grid_waves, c->direct_waves, c->run_waves}) * BLOCK * sizeof(uint4);
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

/// box_quick.h's slack constant for a converted tree: its margins are sized by the largest plane coordinate of the tree — the root's,
/// since every box was checked to lie inside its parent's — and its lemmas want planes that are normal numbers or zero. +inf (no quick
/// box answers) for a tree with irregular or non-bounding boxes or a subnormal plane.
float tree_slack(const Converter &cv, const Converter::Child &root) {
    if (cv.irregular || cv.disorderly || cv.subnormal) return __builtin_inff();
    float pmax = 0;
    for (int k = 0; k < 3; k++) pmax = std::max(pmax, std::max(std::fabs(root.bmin[k]), std::fabs(root.bmax[k])));
    return gq_slack_of_tree(pmax);
}

Scene scene_of(const gpuart_hip_ctx *c) {
    Scene s;
    s.recs = c->d_recs;
    memcpy(s.root_min, c->root_min, 12); memcpy(s.root_max, c->root_max, 12);
    s.root_ref = c->root_ref;
    s.exact_boxes = c->exact_boxes;
    s.box_slack = c->box_slack;
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

template <class T, class A>
int upload_vec(gpuart_hip_ctx *c, T *&dst, const std::vector<T, A> &v) {
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
static int comm_drop(gpuart_hip_ctx *c);

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
    c->direct_waves = c->num_cus * env_u32("GPUART_HIP_DIRECT_WAVES_PER_CU", 16, 1, 32);
    c->shade_waves = c->num_cus * env_u32("GPUART_HIP_SHADE_WAVES_PER_CU", 24, 1, 64);
    c->run_waves = c->num_cus * env_u32("GPUART_HIP_RUN_WAVES_PER_CU", 4 * GD_RUN_WAVES, 1, 32);
    c->tune.chunk = env_u32("GPUART_HIP_CHUNK", 128, 16, 4096);
    c->chunk_from_env = getenv("GPUART_HIP_CHUNK") != nullptr;
    c->tune.refill_lanes = env_u32("GPUART_HIP_REFILL_LANES", 16, 1, 64);
    c->tune.leaf_lanes = env_u32("GPUART_HIP_LEAF_LANES", 16, 1, 64);
    c->tune.leaf_share = env_u32("GPUART_HIP_LEAF_SHARE", 3, 1, 64);
    c->tune.xcd_queues = env_u32("GPUART_HIP_XCD_QUEUES", 0, 0, 1);
    c->order_auto = env_u32("GPUART_HIP_TILE_ORDER", 1, 0, 1) != 0;
    c->nearest_min_prims = getenv("GPUART_HIP_NEAREST_MIN_PRIMS") ? env_u32("GPUART_HIP_NEAREST_MIN_PRIMS", 1024, 0, 0x7fffffff) : 0xffffffffu;
    c->quick_boxes = env_u32("GPUART_HIP_QUICK_BOXES", 1, 0, 1);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(GPUART_HIP_ERR_DEVICE, "hipStreamCreate failed"); }
    c->lanes.resize(env_u32("GPUART_HIP_PASSES_IN_FLIGHT", 8, 1, 32));
    c->plan.lanes_total = (uint32_t)c->lanes.size();
    c->plan.batch_limit = env_u32("GPUART_HIP_MAX_BATCH", MAX_BATCH, 1, MAX_BATCH);
    c->plan.batch_paths = (size_t)env_u32("GPUART_HIP_BATCH_MPATHS", 16, 1, 256) << 20;
    c->plan.plan_run_factor = env_u32("GPUART_HIP_PLAN_RUN_PERCENT", 0, 0, 1000) / 100.0;  // 0: as many equal runs as lanes (run_planner.h)
    c->lean_kernels = env_u32("GPUART_HIP_LEAN_KERNELS", 1, 0, 1) != 0;
    c->plan.min_run_paths = (size_t)env_u32("GPUART_HIP_MIN_RUN_KPATHS", 2048, 64, 65536) << 10;
    c->plan.lane_budget = (size_t)env_u32("GPUART_HIP_LANE_BUDGET_MB", 32768, 64, 262144) << 20;
    c->plan.small_paths = (size_t)env_u32("GPUART_HIP_SMALL_KPATHS", 8500, 0, 1 << 20) << 10;
    c->gather_timeout_ms = env_u32("GPUART_HIP_GATHER_TIMEOUT_MS", 60000, 0, 3600000);
    for (auto &l : c->lanes) {
        if (hipStreamCreateWithFlags(&l.main, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&l.ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&l.ev_free, hipEventDisableTiming) != hipSuccess) {
            gpuart_hip_destroy(c);
            return fail(GPUART_HIP_ERR_DEVICE, "stream / event creation failed");
        }
    }
    if (hipMalloc(&c->d_counters, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), c->stream) != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c; return fail(GPUART_HIP_ERR_DEVICE, "counter allocation failed");
    }
    *out = c;
    return 0;
}

int gpuart_hip_destroy(gpuart_hip_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    c->pend_seeds.clear();
    if (c->abandoned || bounded_ns::stuck().load()) {
        // the primary stream may hold transfers whose peer never came: one more bounded look, then the context is LEAKED rather
        // than waited for (its memory goes with the process)
        if (wait_stream(c, std::min<uint32_t>(c->gather_timeout_ms ? c->gather_timeout_ms : 2000u, 2000u), "gpuart_hip_destroy"))
            return fail(GPUART_HIP_ERR_TIMEOUT, "gpuart_hip_destroy: the context's stream never drained (a gather was abandoned); the context is leaked, not freed");
        c->abandoned = false;
    }
    (void)drain(c);
    for (auto &t : c->pending) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto &t : c->free_events) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto &l : c->lanes) {
        if (l.ev_done) (void)hipEventDestroy(l.ev_done);
        if (l.ev_free) (void)hipEventDestroy(l.ev_free);
        void *lp[] = {l.pathmem, l.spill_main, l.pb.counters, l.run_cursor};
        for (void *p : lp) if (p) (void)hipFree(p);
        if (l.main) (void)hipStreamDestroy(l.main);
    }
    // A communicator call that never returned (bounded.h) may still be using the communicator and the buffers it was given:
    // they are left alone, and so is the communicator (the process is about to end; nothing is waited for a second time).
    const bool comm_stuck = bounded_ns::stuck().load();
    if (c->comm && !comm_stuck) (void)comm_drop(c);
    void *ptrs[] = {c->d_recs, c->d_prims, c->d_spill, c->d_direct, c->d_accum, c->d_counters, c->d_scratch, c->d_cursor,
                    c->d_tile_order[0], c->d_tile_order[1], c->d_tile_cost};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    void *comm_ptrs[] = {c->d_send, c->d_stage, c->d_hello};
    for (void *p : comm_ptrs) if (p && !comm_stuck) (void)hipFree(p);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->h_hello && !comm_stuck) (void)hipHostFree(c->h_hello);  // (after drain: no copy into it is queued any more)
    if (c->h_frame && !comm_stuck) (void)hipHostFree(c->h_frame);
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
    if (const char *e = getenv("GPUART_HIP_TOP_DEPTH")) cv.top_depth = (uint32_t)atoi(e);
    Converter::Child root;
    unsigned upload_threads = nquads >= (1u << 20) ? 8u : 1u;  // the fill pass of a big tree on several threads (disjoint writes)
    if (const char *e = getenv("GPUART_HIP_UPLOAD_THREADS")) upload_threads = (unsigned)std::min(64, std::max(1, atoi(e)));
    if (!cv.convert(root, upload_threads))
        return fail(GPUART_HIP_ERR_ARG, "malformed compiled BVH: " + cv.err);
    int r;
    if ((r = gpuart_hip_flush(c))) return r;
    if ((r = drain(c))) return r;
    if ((r = upload_vec(c, c->d_recs, cv.recs))) return r;
    if ((r = upload_vec(c, c->d_prims, cv.prims))) return r;
    memcpy(c->root_min, root.bmin, 12); memcpy(c->root_max, root.bmax, 12);
    c->root_ref = root.ref;
    c->exact_boxes = (cv.irregular || cv.disorderly) ? 1u : 0u;
    c->box_slack = c->quick_boxes ? tree_slack(cv, root) : __builtin_inff();
    c->type_mask = cv.type_mask;
    c->n_nodes = cv.num_nodes;
    c->n_prims = cv.prims.size() / 3;
    c->ref_order = (!c->exact_boxes && c->n_prims < c->nearest_min_prims) ? 1u : 0u;
    c->max_depth = cv.max_depth;
    c->scene_bytes = cv.recs.size() * 16 + cv.prims.size() * 16;
    if ((r = ensure_spill(c))) return r;
    c->have_scene = true;
    c->order_stale = true;
    return 0;
}

int gpuart_hip_set_camera(gpuart_hip_ctx *c, const float pos[3], const float bl[3], const float dh[3], const float dv[3]) {
    if (!c || !pos || !bl || !dh || !dv) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }  // batched passes were requested with the old camera
    c->order_stale = true;  // the blocks' cost is counted again by the next run (run_frame)
    memcpy(c->frame.cam_pos, pos, 12); memcpy(c->frame.bottom_left, bl, 12);
    memcpy(c->frame.delta_horz, dh, 12); memcpy(c->frame.delta_vert, dv, 12);
    c->have_camera = true;
    return 0;
}

static int check_ready(gpuart_hip_ctx *c, const gpuart_params *p) {
    if (!c || !p) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    if (c->empty_share) return 1;  // nothing to render (the callers return success)
    if (!c->have_scene) return fail(GPUART_HIP_ERR_ARG, "no scene uploaded");
    if (!c->have_camera) return fail(GPUART_HIP_ERR_ARG, "no camera set");
    if (!c->plan.tile_pixels) return fail(GPUART_HIP_ERR_ARG, "no frame size set");
    return 0;
}

int gpuart_hip_render_direct(gpuart_hip_ctx *c, const gpuart_params *p) {
    int r = check_ready(c, p);
    if (r) return r > 0 ? 0 : r;
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    HIP_TRY(hipSetDevice(c->device));
    dim3 grid(std::min<uint32_t>(c->grid_waves, c->plan.n_slots / BLOCK));
    Scene sc = scene_of(c);
    TimedLaunch t;
    if ((r = begin_timed(c, t, 0))) return r;
    if (c->plan.mode == 1) {
        k_direct<true><<<grid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, c->plan.n_slots, c->d_direct, c->d_spill, c->d_counters);
    } else if (c->plan.mode == 2) {
        k_direct<false><<<grid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, c->plan.n_slots, c->d_direct, c->d_spill, c->d_counters);
    } else {
        // fast mode: persistent lanes, one pixel at a time per lane (kernels_pipeline.h)
        if (!c->d_cursor) HIP_TRY(hipMalloc(&c->d_cursor, 64));
        HIP_TRY(hipMemsetAsync(c->d_cursor, 0, sizeof(uint32_t), c->stream));
        const dim3 pgrid(c->direct_waves);  // alone on the GPU: 16 waves per CU measured best
        // The frame is one launch: its tail is the last chunk a wave takes from the cursor (128 pixels are two rounds of two
        // dependent queries). Small frames take 64-pixel chunks — 1080p 0.92 -> 0.75 ms per frame; 4K stays at 128 (1.74 against
        // 1.85): fewer cursor atomics, longer coherent runs (gpurun_out/direct_chunk.txt; 32 and 16 are far worse: 1.0 / 1.7 ms)
        TraceTuning dtune = c->tune;
        if (!c->chunk_from_env && c->plan.n_slots / ((size_t)c->direct_waves * 8) < 128) dtune.chunk = 64;
        const bool flat_only = c->lean_kernels && !c->exact_boxes && (c->type_mask & ~(uint32_t)GD_FLAT_TYPES) == 0;
        const bool round_only = c->lean_kernels && !c->exact_boxes && !flat_only && (c->type_mask & ~(uint32_t)GD_ROUND_TYPES) == 0;
        if (c->exact_boxes) k_direct_persistent<GD_ALL_TYPES | GD_EXACT_BOXES><<<pgrid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, c->plan.n_slots, c->d_direct, c->d_spill, c->d_cursor, dtune);
        else if (flat_only) k_direct_persistent<GD_FLAT_TYPES><<<pgrid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, c->plan.n_slots, c->d_direct, c->d_spill, c->d_cursor, dtune);
        else if (round_only) k_direct_persistent<GD_ROUND_TYPES><<<pgrid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, c->plan.n_slots, c->d_direct, c->d_spill, c->d_cursor, dtune);
        else k_direct_persistent<GD_ALL_TYPES><<<pgrid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, c->plan.n_slots, c->d_direct, c->d_spill, c->d_cursor, dtune);
    }
    HIP_TRY(hipGetLastError());
    return end_timed(c, t);
}

int gpuart_hip_pt_plan(gpuart_hip_ctx *c, uint32_t passes) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    c->plan.set_plan(passes);
    return 0;
}

int gpuart_hip_pt_reset(gpuart_hip_ctx *c) {
    if (c && c->empty_share) return 0;
    if (!c || !c->plan.tile_pixels) return fail(GPUART_HIP_ERR_ARG, "no frame size set");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    // passes still in flight belong to the accumulation that is being discarded: let them finish first
    int r = drain(c);
    if (r) return r;
    c->plan.plan_done = 0;  // a new accumulation: the planned sequence, if any, begins again
    HIP_TRY(hipMemsetAsync(c->d_accum, 0, c->plan.tile_pixels * sizeof(float4), c->stream));
    return 0;
}

/// Upper bound on the number of segments any path can have. colorWeight is multiplied per segment by the albedo
/// of the primitive type that was hit (path_tracing.glsl:123-126,204) and the loop stops as soon as ANY channel
/// is <= minWeight, so channel c survives at most as long as (largest albedo.c among the primitive types present
/// in the scene)^n > minWeight; the bound is the smallest such n over the channels. The bound may only ever be too
/// LARGE (a launch that finds an empty queue costs microseconds; a missing one would leave paths uncommitted), so the
/// comparison is relaxed by 1e-4: fp32 products of mixed albedos can round differently from the powers taken here.
/// k_shade additionally commits a path whose next segment would lie beyond the launched ones (belt and braces).
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
        while (k < n && w > p->minWeight * 0.9999f) { w *= a; k++; }
        bound = std::min(bound, k);
    }
    return std::max<uint32_t>(bound, 1);
}

extern "C++" {
namespace {
/// The frame as the kernels of ONE run see it: with the birth order of its paths (one order for all launches of the run: path
/// slots keep their pixels from launch to launch) and, if this run is to count the cost of the tile's blocks, the counters.
/// The library orders by itself only where that was measured to pay (profiles/r04/birth_order.txt): runs of ONE pass through k_run
/// — the frame-by-frame use, where the run's tail is a third of its time. Longer runs keep the row-major order: most expensive
/// first shortens their (relatively smaller) tail too but puts all heavy blocks in flight at once, which costs the main phase more.
int run_frame(gpuart_hip_ctx *c, Frame &f, bool &gathers, bool single_pass_run) {
    f = c->frame;
    gathers = false;
    if (c->order_from_hook || !c->order_auto || !single_pass_run || (c->plan.mode != 0 && c->plan.mode != 5)) return 0;
    if (c->order_sorting && hipEventQuery(c->ev_order) == hipSuccess) {  // (hipErrorNotReady: keep the order in use a little longer)
        c->order_cur = c->order_next;
        c->order_sorting = false;
    }
    (void)hipGetLastError();
    f.tile_order = c->order_cur >= 0 ? c->d_tile_order[c->order_cur] : nullptr;
    if (!c->order_sorting && c->order_stale) {
        const size_t tiles = (size_t)((c->frame.tw + 7) / 8) * ((c->frame.th + 7) / 8);
        if (!c->d_tile_cost) {
            HIP_TRY(hipMalloc(&c->d_tile_cost, tiles * sizeof(uint32_t)));
            HIP_TRY(hipMemset(c->d_tile_cost, 0, tiles * sizeof(uint32_t)));  // (k_tile_order zeroes what it has read)
            for (auto &o : c->d_tile_order) if (!o) HIP_TRY(hipMalloc(&o, tiles * sizeof(uint32_t)));
            if (!c->ev_order) HIP_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
        }
        f.tile_cost = c->d_tile_cost;
        gathers = true;
    }
    return 0;
}

/// Behind a gathering run on its lane's stream: the sort into the order buffer that no run uses — none launched so far may still
/// read it (the lanes' ev_done), and the runs to come take it only once ev_order has been seen complete (run_frame).
int sort_tile_order(gpuart_hip_ctx *c, PassLane &l) {
    for (auto &o : c->lanes) if (&o != &l && o.used) HIP_TRY(hipStreamWaitEvent(l.main, o.ev_done, 0));
    c->order_next = c->order_cur == 0 ? 1 : 0;
    const size_t tiles = (size_t)((c->frame.tw + 7) / 8) * ((c->frame.th + 7) / 8);
    k_tile_order<<<1, TO_THREADS, 0, l.main>>>(c->d_tile_cost, c->d_tile_order[c->order_next], (uint32_t)tiles);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_order, l.main));
    c->order_sorting = true;
    c->order_stale = false;
    return 0;
}

/// The collected passes [first, first + count) as ONE persistent kernel per path of the pass (k_run, kernel_run.h) on
/// pass lane `l`; then their colour planes are added to the accumulator in pass order on the primary stream.
int launch_run_persistent(gpuart_hip_ctx *c, PassLane &l, size_t first, size_t count) {
    int r;
    const gpuart_params *p = &c->pend_params;
    const int npaths = c->pend_npaths;
    Scene sc = scene_of(c);
    SeedBatch seeds{};
    for (size_t k = 0; k < count; k++) seeds.seed[k] = c->pend_seeds[first + k];
    if (!l.run_cursor) HIP_TRY(hipMalloc(&l.run_cursor, 64));
    const PathBuffers &b = l.pb;
    const bool flat_only = c->lean_kernels && !c->exact_boxes && (c->type_mask & ~(uint32_t)GD_FLAT_TYPES) == 0;  // triangle meshes + discs
    const bool round_only = c->lean_kernels && !c->exact_boxes && !flat_only && (c->type_mask & ~(uint32_t)GD_ROUND_TYPES) == 0;  // spheres + discs
    const bool exact = c->exact_boxes != 0;  // a tree with irregular boxes: the kernel variants with comparison-form box tests
    // k_run's list entries carry the path slot in RUN_SLOT's bits beside their flags
    // (the LARGEST slot index, n_slots x batch - 1, must fit: a run of exactly 2^28 slots does; the planner never plans a longer one
    //  for mode 0 — run_planner.h RUN_KERNEL_SLOTS — so only modes 1 / 4 / 5 on a tile beyond 2^28 pixels can get here)
    if ((uint64_t)b.n_slots * b.batch > (uint64_t)RUN_SLOT + 1)
        return fail(GPUART_HIP_ERR_ARG, "modes 1, 4 and 5 run through k_run, whose list entries address 2^28 path slots per run; this tile has " +
                    std::to_string((uint64_t)b.n_slots * b.batch) + " (use mode 0 or 3, or a smaller tile)");
    const uint32_t chunks = b.n_slots * b.batch / BLOCK;
    const dim3 grid(std::min<uint32_t>(c->run_waves, std::max<uint32_t>(1, chunks)));
    TimedLaunch t;
    Frame fr;
    bool gathers;
    if ((r = run_frame(c, fr, gathers, count == 1))) return r;
    if (l.used) HIP_TRY(hipStreamWaitEvent(l.main, l.ev_free, 0));  // the lane's previous run has been accumulated
    if ((r = begin_timed(c, t, 0, l.main))) return r;
    for (int j = 0; j < npaths; j++) {
        TimedLaunch tt;
        HIP_TRY(hipMemsetAsync(l.run_cursor, 0, sizeof(uint32_t), l.main));
        if (c->timing_level >= 2 && (r = begin_timed(c, tt, 1, l.main))) return r;
#define GD_LAUNCH_RUN(C, R, T) k_run<C, R, T><<<grid, BLOCK, 0, l.main>>>(sc, fr, *p, seeds, b, j, npaths, l.passcolor, l.spill_main, c->d_counters, c->tune, l.run_cursor)
#define GD_LAUNCH_RUN_ORD(C, T) do { if (c->ref_order) GD_LAUNCH_RUN(C, false, (T) | GD_REF_ORDER); else GD_LAUNCH_RUN(C, false, T); } while (0)
        if (c->plan.mode == 1 && exact) GD_LAUNCH_RUN(true, true, GD_ALL_TYPES | GD_EXACT_BOXES);
        else if (c->plan.mode == 1) GD_LAUNCH_RUN(true, true, GD_ALL_TYPES);
        else if (c->plan.mode == 4 && exact) GD_LAUNCH_RUN(true, false, GD_ALL_TYPES | GD_EXACT_BOXES);
        else if (c->plan.mode == 4 && flat_only) GD_LAUNCH_RUN_ORD(true, GD_FLAT_TYPES);
        else if (c->plan.mode == 4) GD_LAUNCH_RUN_ORD(true, GD_ALL_TYPES);
        else if (exact) GD_LAUNCH_RUN(false, false, GD_ALL_TYPES | GD_EXACT_BOXES);
        else if (flat_only) GD_LAUNCH_RUN_ORD(false, GD_FLAT_TYPES);
        else if (round_only) GD_LAUNCH_RUN_ORD(false, GD_ROUND_TYPES);
        else GD_LAUNCH_RUN_ORD(false, GD_ALL_TYPES);
#undef GD_LAUNCH_RUN_ORD
#undef GD_LAUNCH_RUN
        if (c->timing_level >= 2 && (r = end_timed(c, tt, l.main))) return r;
    }
    if ((r = end_timed(c, t, l.main))) return r;
    HIP_TRY(hipEventRecord(l.ev_done, l.main));
    HIP_TRY(hipStreamWaitEvent(c->stream, l.ev_done, 0));
    k_accumulate<<<dim3((unsigned)((c->plan.tile_pixels + 255) / 256)), 256, 0, c->stream>>>(c->d_accum, l.passcolor, c->plan.tile_pixels, b.batch);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(l.ev_free, c->stream));
    if (gathers && (r = sort_tile_order(c, l))) return r;
    l.used = true;
    return 0;
}

/// One run of the launch-per-stage wavefront pipeline (mode 3) for the pending passes [first, first + count).
/// One run of the wavefront pipeline for the pending passes [first, first + count) on the next pass lane.
int launch_run(gpuart_hip_ctx *c, size_t first, size_t count) {
    int r;
    const gpuart_params *p = &c->pend_params;
    const int npaths = c->pend_npaths;
    Scene sc = scene_of(c);
    TimedLaunch t;
    // a lane's path buffers hold max_batch passes (realloc_tile): a longer run would write past them (run_planner.h)
    if (const char *bad = c->plan.check(count)) return fail(GPUART_HIP_ERR_DEVICE, std::string("internal: ") + bad);
    PassLane &l = c->lanes[c->next_lane];
    c->next_lane = (c->next_lane + 1) % c->plan.lanes_in_use;
    l.pb.batch = (uint32_t)count;
    if (c->plan.uses_run_kernel(count)) return launch_run_persistent(c, l, first, count);
    const uint32_t nseg = segment_bound(c, p);
    SeedBatch seeds{};
    for (size_t k = 0; k < count; k++) seeds.seed[k] = c->pend_seeds[first + k];
    if ((r = ensure_segment_counters(c, l, nseg))) return r;
    const PathBuffers &b = l.pb;
    const bool flat_only = c->lean_kernels && !c->exact_boxes && (c->type_mask & ~(uint32_t)GD_FLAT_TYPES) == 0;  // triangle meshes + discs
    const bool round_only = c->lean_kernels && !c->exact_boxes && !flat_only && (c->type_mask & ~(uint32_t)GD_ROUND_TYPES) == 0;  // spheres + discs
    const bool detail = c->timing_level >= 2;
    const dim3 pgrid(c->grid_waves);
    const dim3 sgrid(std::min<uint32_t>(c->shade_waves, b.n_slots * b.batch / BLOCK));  // k_gen / k_shade: grid-stride loops
    int j_cur = 0;
    Frame fr;
    bool gathers;
    if ((r = run_frame(c, fr, gathers, false))) return r;  // (an order set through gpuart_hip_test_tile_order applies here too)
    if (l.used) HIP_TRY(hipStreamWaitEvent(l.main, l.ev_free, 0));  // the lane's previous pass has been accumulated
    if ((r = begin_timed(c, t, 0, l.main))) return r;
    // one BVH-query launch: closest-hit queries of segment seg_c and / or Sun-shadow queries of segment seg_s
    auto trace = [&](int seg_c, int seg_s) -> int {
        TimedLaunch tt;
        int rr;
        if (detail && (rr = begin_timed(c, tt, 1, l.main))) return rr;
#define GD_LAUNCH_TRACE(T) k_trace<false, T><<<pgrid, BLOCK, 0, l.main>>>(sc, fr, *p, b, seg_c, seg_s, 1, j_cur, npaths, l.passcolor, l.spill_main, c->d_counters, c->tune)
#define GD_LAUNCH_TRACE_ORD(T) do { if (c->ref_order) GD_LAUNCH_TRACE((T) | GD_REF_ORDER); else GD_LAUNCH_TRACE(T); } while (0)
        if (c->exact_boxes) GD_LAUNCH_TRACE(GD_ALL_TYPES | GD_EXACT_BOXES);
        else if (flat_only) GD_LAUNCH_TRACE_ORD(GD_FLAT_TYPES);
        else if (round_only) GD_LAUNCH_TRACE_ORD(GD_ROUND_TYPES);
        else GD_LAUNCH_TRACE_ORD(GD_ALL_TYPES);
#undef GD_LAUNCH_TRACE_ORD
#undef GD_LAUNCH_TRACE
        if (detail && (rr = end_timed(c, tt, l.main))) return rr;
        return 0;
    };
    for (int j = 0; j < npaths; j++) {
        j_cur = j;
        HIP_TRY(hipMemsetAsync(b.counters, 0, 4 * ((size_t)nseg + 1) * sizeof(uint32_t), l.main));
        if (c->tune.xcd_queues) HIP_TRY(hipMemsetAsync(b.xcd_cursors, 0, 16 * ((size_t)nseg + 1) * sizeof(uint32_t), l.main));
        k_gen<<<sgrid, BLOCK, 0, l.main>>>(fr, *p, seeds, j, npaths, b, l.passcolor);
        if (nseg && (r = trace(0, -1))) return r;
        for (uint32_t seg = 0; seg < nseg; seg++) {
            k_shade<false><<<sgrid, BLOCK, 0, l.main>>>(sc, fr, *p, seeds, b, (int)seg, (int)nseg, j, npaths, l.passcolor, c->d_counters);
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
    k_accumulate<<<dim3((unsigned)((c->plan.tile_pixels + 255) / 256)), 256, 0, c->stream>>>(c->d_accum, l.passcolor, c->plan.tile_pixels, b.batch);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(l.ev_free, c->stream));
    if (gathers && (r = sort_tile_order(c, l))) return r;
    l.used = true;
    return 0;
}
}  // namespace
}  // extern "C++"

int gpuart_hip_flush(gpuart_hip_ctx *c) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    const size_t pending = c->pend_seeds.size();
    if (!pending) return 0;
    HIP_TRY(hipSetDevice(c->device));
    // What is still pending when something observes or changes state (read-back, finish, ...) is split into runs
    // (RunPlanner::on_flush)
    c->plan.pending = pending;
    int r = 0;
    size_t first = 0;
    for (size_t count : c->plan.on_flush()) {
        if ((r = launch_run(c, first, count))) break;
        first += count;
    }
    c->pend_seeds.clear();
    return r;
}


int gpuart_hip_pt_pass(gpuart_hip_ctx *c, const gpuart_params *p, const float randSeed[4], int npaths) {
    int r = check_ready(c, p);
    if (r) return r > 0 ? 0 : r;
    if (!randSeed || npaths < 0) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    if (p->maxSegments > GPUART_HIP_MAX_SEGMENTS) return fail(GPUART_HIP_ERR_ARG, "maxSegments exceeds GPUART_HIP_MAX_SEGMENTS");
    if (npaths == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    const float4 seed = make_float4(randSeed[0], randSeed[1], randSeed[2], randSeed[3]);
    if (c->plan.mode == 2) {  // megakernel: the whole path in one thread, on the primary stream (ablation / cross-check)
        if ((r = gpuart_hip_flush(c))) return r;
        Scene sc = scene_of(c);
        TimedLaunch t;
        if ((r = begin_timed(c, t, 0))) return r;
        dim3 grid(std::min<uint32_t>(c->grid_waves, c->plan.n_slots / BLOCK));
        k_pt_mega<false><<<grid, BLOCK, 0, c->stream>>>(sc, c->frame, *p, seed, npaths, c->plan.n_slots, c->d_accum, c->d_spill, c->d_counters);
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
    c->plan.pending = c->pend_seeds.size() - 1;
    if (const size_t count = c->plan.on_pass()) {
        r = launch_run(c, 0, count);
        c->pend_seeds.clear();
        return r;
    }
    return 0;
}

int gpuart_hip_export(gpuart_hip_ctx *c, int which, void *rgba_device, float divide_by) {
    if (!c || !rgba_device || (which != 0 && which != 1) || !c->plan.tile_pixels) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    const float4 *src = which == 0 ? c->d_direct : c->d_accum;
    if (!(divide_by > 0)) divide_by = 1.0f;
    size_t n = c->plan.tile_pixels;
    k_scale_copy<<<dim3((unsigned)((n + 255) / 256)), 256, 0, c->stream>>>(src, (float4 *)rgba_device, n, divide_by);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gpuart_hip_read(gpuart_hip_ctx *c, int which, float *rgba_host, float divide_by) {
    if (!c || !rgba_host || (which != 0 && which != 1) || !c->plan.tile_pixels) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    size_t bytes = c->plan.tile_pixels * sizeof(float4);
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
    if (!c || !rgba_host || which != 1 || !c->plan.tile_pixels) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    HIP_TRY(hipMemcpyAsync(c->d_accum, rgba_host, c->plan.tile_pixels * sizeof(float4), hipMemcpyHostToDevice, c->stream));
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
    if (!c || mode < 0 || mode > 5) return fail(GPUART_HIP_ERR_ARG, "bad mode");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    int r = drain(c);  // modes use different streams; keep their passes ordered
    if (r) return r;
    c->plan.mode = mode;
    c->plan.plan();
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
    unsigned long long h[16];
    int rr = drain(c);
    if (rr) return rr;
    HIP_TRY(hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (out) {
        out->rays = h[0]; out->nodes = h[1];
        for (int k = 0; k < 4; k++) out->prim_tests[k] = h[2 + k];
        out->segments = h[6];
        out->box_steps = h[7];
        out->box_steps_top = h[8];
        out->rewalks = h[9];
    }
    if (reset) HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof h, c->stream));
    return 0;
}

int gpuart_hip_set_nearest_first(gpuart_hip_ctx *c, uint32_t min_prims) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }  // (passes collected so far were requested under the old choice; both give... the reference's image or, opted in, the walk's)
    c->nearest_min_prims = min_prims;
    if (c->have_scene) c->ref_order = (!c->exact_boxes && c->n_prims < c->nearest_min_prims) ? 1u : 0u;
    return 0;
}

int gpuart_hip_scene_order(gpuart_hip_ctx *c, int *order) {
    if (!c || !order || !c->have_scene) return fail(GPUART_HIP_ERR_ARG, "no scene uploaded");
    *order = c->exact_boxes ? 2 : c->ref_order ? 1 : 0;
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

// ---- shares and the multi-GPU gather (gather.h) ---------------------------------------------------------------------
int gpuart_hip_set_share(gpuart_hip_ctx *c, const gpuart_tile_geom *g) {
    if (!c || !g || !geom_ok(*g)) return fail(GPUART_HIP_ERR_ARG, "bad share");
    if (g->W != c->frame.W || g->H != c->frame.H) return fail(GPUART_HIP_ERR_ARG, "share of another frame size (call gpuart_hip_resize first)");
    if (g->th) return gpuart_hip_set_tile_interleaved(c, g->x0, g->y0, g->tw, g->th, g->band_rows, g->band_stride);
    // An empty share (more ranks than bands): the context renders nothing — render calls succeed and do nothing — and still
    // takes part in gpuart_hip_gather, where it sends nothing.
    HIP_TRY(hipSetDevice(c->device));
    c->frame.x0 = g->x0; c->frame.y0 = g->y0; c->frame.tw = g->tw; c->frame.th = 0;
    c->frame.band_rows = g->band_rows; c->frame.band_stride = g->band_stride;
    int r = realloc_tile(c);
    c->empty_share = r == 0;
    return r;
}

int gpuart_hip_get_share(gpuart_hip_ctx *c, gpuart_tile_geom *g) {
    if (!c || !g || !c->frame.W) return fail(GPUART_HIP_ERR_ARG, "no frame size set");
    g->W = c->frame.W; g->H = c->frame.H; g->x0 = c->frame.x0; g->y0 = c->frame.y0; g->tw = c->frame.tw; g->th = c->frame.th;
    g->band_rows = c->frame.band_rows; g->band_stride = c->frame.band_stride;
    return 0;
}

#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t e_ = (expr);                                                                       \
        if (e_ != ncclSuccess)                                                                          \
            return fail(GPUART_HIP_ERR_DEVICE, std::string(#expr) + ": " + rccl()->GetErrorString(e_)); \
    } while (0)

static int need_rccl() {
    Rccl *r = rccl();
    if (!r->err.empty()) return fail(GPUART_HIP_ERR_DEVICE, r->err);
    return 0;
}

int gpuart_hip_comm_unique_id(void *id128) {
    if (!id128) return fail(GPUART_HIP_ERR_ARG, "id == NULL");
    int r = need_rccl();
    if (r) return r;
    static_assert(sizeof(ncclUniqueId) == GPUART_HIP_UNIQUE_ID_BYTES, "unique id size");
    NCCL_TRY(rccl()->GetUniqueId((ncclUniqueId *)id128));
    return 0;
}

/// An RCCL-facing entry point in a process where an earlier RCCL call never returned (bounded.h): refused at once.
static int comm_layer_ok() {
    if (bounded_ns::stuck().load())
        return fail(GPUART_HIP_ERR_TIMEOUT, "the communicator layer of this process is out of service: " + bounded_ns::stuck_in() + " never returned");
    return 0;
}

/// A bounded RCCL call's outcome as this library's status (the message is set on the CALLING thread).
static int comm_outcome(const bounded_ns::Outcome &o, const char *what) {
    if (o.timed_out) return fail(GPUART_HIP_ERR_TIMEOUT, o.detail);
    if (o.rc != 0) return fail(GPUART_HIP_ERR_DEVICE, std::string(what) + ": " + (o.detail.empty() ? std::string("failed") : o.detail));
    return 0;
}

/// Leaves the context's communicator. ncclCommDestroy waits for the communicator's outstanding work and, for a communicator of
/// several ranks, for its proxy threads — it is one of the calls that had no bound: it runs under bounded() now.
/// A destroy that does not return leaves the handle alone (the context forgets it) and marks the layer stuck.
static int comm_drop(gpuart_hip_ctx *c) {
    int r = 0;
    if (c->comm && c->comm_owned && !bounded_ns::stuck().load()) {
        const ncclComm_t comm = c->comm;
        const int device = c->device;
        const bounded_ns::Outcome o = bounded_ns::bounded("ncclCommDestroy", bounded_ns::comm_timeout_ms(), [comm, device](std::string &d) {
            (void)hipSetDevice(device);
            const ncclResult_t e = rccl()->CommDestroy(comm);
            if (e != ncclSuccess) d = rccl()->GetErrorString(e);
            return (int)e;
        });
        r = comm_outcome(o, "ncclCommDestroy");
    }
    c->comm = nullptr; c->comm_owned = false; c->comm_nranks = 0; c->comm_rank = 0; c->comm_group = 0;
    return r;
}

/// Identity of a communicator for gpuart_hip_gather_all: process-unique serials for _comm_init_all / _comm_attach, a hash of the
/// unique id for _comm_init (the same on every rank that was given that id).
static uint64_t next_comm_group() {
    static std::atomic<uint64_t> serial{1};
    return serial.fetch_add(1);
}

int gpuart_hip_comm_init(gpuart_hip_ctx *c, int nranks, int rank, const void *id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    int r = comm_layer_ok();
    if (r || (r = need_rccl())) return r;
    HIP_TRY(hipSetDevice(c->device));
    if ((r = comm_drop(c))) return r;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    // ncclCommInitRank returns when ALL nranks ranks have called it with this id: a rank that never starts leaves the others here
    auto made = std::make_shared<ncclComm_t>(nullptr);
    const int device = c->device;
    const bounded_ns::Outcome o = bounded_ns::bounded("ncclCommInitRank", bounded_ns::comm_timeout_ms(), [made, nranks, id, rank, device](std::string &d) {
        (void)hipSetDevice(device);
        const ncclResult_t e = rccl()->CommInitRank(made.get(), nranks, id, rank);
        if (e != ncclSuccess) d = rccl()->GetErrorString(e);
        return (int)e;
    });
    if ((r = comm_outcome(o, "ncclCommInitRank"))) return r;
    c->comm = *made;
    c->comm_owned = true; c->comm_nranks = nranks; c->comm_rank = rank;
    uint64_t h = 1469598103934665603ull;  // FNV-1a of the 128-byte id; bit 63 set: never a serial
    for (size_t k = 0; k < sizeof id; k++) h = (h ^ ((const unsigned char *)&id)[k]) * 1099511628211ull;
    c->comm_group = h | (1ull << 63);
    return 0;
}

int gpuart_hip_comm_attach(gpuart_hip_ctx *c, void *nccl_comm, int nranks, int rank) {
    if (!c || !nccl_comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    int r = comm_layer_ok();
    if (r || (r = need_rccl())) return r;
    if ((r = comm_drop(c))) return r;
    c->comm = (ncclComm_t)nccl_comm; c->comm_owned = false; c->comm_nranks = nranks; c->comm_rank = rank;
    c->comm_group = 0;  // the caller's communicator: which handles belong together is the caller's to know
    return 0;
}

int gpuart_hip_comm_init_all(gpuart_hip_ctx *const *ctxs, int n) {
    if (!ctxs || n < 1) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    int r = comm_layer_ok();
    if (r || (r = need_rccl())) return r;
    // RCCL refuses two ranks of a communicator on one device. An in-process stand-in (GPUART_HIP_RCCL_LIBRARY, tests/stubs/rccl_stub.cpp)
    // does not: with GPUART_HIP_TEST_SHARED_DEVICE=1 beside it the N > 1 gather runs on a box with one GPU. Test-only, both variables.
    const char *shared = getenv("GPUART_HIP_TEST_SHARED_DEVICE");
    const bool one_device_ok = getenv("GPUART_HIP_RCCL_LIBRARY") && shared && shared[0] == '1';
    std::vector<int> devs(n);
    for (int k = 0; k < n; k++) {
        if (!ctxs[k]) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
        devs[k] = ctxs[k]->device;
        for (int m = 0; m < k; m++) {
            if (ctxs[m] == ctxs[k]) return fail(GPUART_HIP_ERR_ARG, "one context given twice");
            if (devs[m] == devs[k] && !one_device_ok) return fail(GPUART_HIP_ERR_ARG, "two contexts of one communicator on the same device");
        }
    }
    // a context leaves its previous communicator first (with its work drained: nothing of that communicator is in flight)
    for (int k = 0; k < n; k++) {
        if (!ctxs[k]->comm) continue;
        HIP_TRY(hipSetDevice(ctxs[k]->device));
        if ((r = drain(ctxs[k]))) return r;
        if ((r = comm_drop(ctxs[k]))) return r;
    }
    // ncclCommInitAll bootstraps n ranks inside this process (one thread per rank inside RCCL, topology discovery, ring set-up):
    // it has no timeout of its own. Under bounded(): the handles live in a block the parked thread shares.
    auto made = std::make_shared<std::vector<ncclComm_t>>((size_t)n, nullptr);
    const bounded_ns::Outcome o = bounded_ns::bounded("ncclCommInitAll", bounded_ns::comm_timeout_ms(), [made, devs, n](std::string &d) {
        const ncclResult_t e = rccl()->CommInitAll(made->data(), n, devs.data());
        if (e != ncclSuccess) d = rccl()->GetErrorString(e);
        return (int)e;
    });
    if ((r = comm_outcome(o, "ncclCommInitAll"))) return r;
    const uint64_t group = next_comm_group();
    for (int k = 0; k < n; k++) {
        ctxs[k]->comm = (*made)[(size_t)k]; ctxs[k]->comm_owned = true; ctxs[k]->comm_nranks = n; ctxs[k]->comm_rank = k;
        ctxs[k]->comm_group = group;
    }
    return 0;
}

int gpuart_hip_comm_destroy(gpuart_hip_ctx *c) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    if (c->comm) {
        HIP_TRY(hipSetDevice(c->device));
        // (a context whose gather gave up may still have that gather's transfers queued: the wait for them is bounded too)
        int r = c->abandoned ? wait_stream(c, c->gather_timeout_ms, "gpuart_hip_comm_destroy: work queued before the communicator can go") : 0;
        if (!r) r = drain(c);
        if (r) return r;
        return comm_drop(c);
    }
    return 0;
}

extern "C++" {
namespace {
int ensure_pixels(float4 *&buf, size_t &have, size_t want) {
    if (have >= want && buf) return 0;
    if (buf) { (void)hipFree(buf); buf = nullptr; have = 0; }
    HIP_TRY(hipMalloc(&buf, std::max<size_t>(1, want) * sizeof(float4)));
    have = want;
    return 0;
}

/// What every rank tells the others before anything is sent: its share and whether it is able to take part.
struct GatherHello {
    gpuart_tile_geom g;
    uint32_t status;  ///< 0: ready; else the rank could not prepare (out of memory, bad arguments): nobody posts a transfer
    uint32_t which, root, pad;
};
static_assert(sizeof(GatherHello) == 48, "all-gathered as 12 words per rank");

/// Everything of a gather that can fail on THIS rank alone, done before the ranks commit to a transfer: pending passes are
/// launched, the (divided) tile is exported into the rank's send buffer, and the root's staging area — one frame, whatever the
/// shares turn out to be — exists. The verdict travels with the share (GatherHello::status), so that a rank that cannot go on
/// makes every rank return an error instead of leaving its peers blocked in a receive.
int gather_prepare(gpuart_hip_ctx *c, int which, float divide_by, int root, GatherHello &h) {
    int r;
    memset(&h, 0, sizeof h);
    h.which = (uint32_t)which; h.root = (uint32_t)root;
    if ((r = gpuart_hip_get_share(c, &h.g))) return r;
    if ((r = gpuart_hip_flush(c))) return r;
    if (c->plan.tile_pixels) {
        if ((r = ensure_pixels(c->d_send, c->send_pixels, c->plan.tile_pixels))) return r;
        if ((r = gpuart_hip_export(c, which, c->d_send, divide_by))) return r;
    }
    if (c->comm_rank == root && (r = ensure_pixels(c->d_stage, c->stage_pixels, (size_t)c->frame.W * c->frame.H))) return r;
    return 0;
}

/// The shares of all ranks must be full-width rows of one frame that cover every row exactly once: rows nobody renders would
/// stay uninitialised in the root's frame, rows rendered twice would depend on the order of the transfers.
int check_shares(const std::vector<GatherHello> &all, int which, int root) {
    const gpuart_tile_geom &g0 = all[0].g;
    // a rank that could not prepare says so before its (zeroed) share is looked at: rank 0's would otherwise read as a bad frame
    for (size_t k = 0; k < all.size(); k++)
        if (all[k].status) return fail(GPUART_HIP_ERR_DEVICE, "gather: rank " + std::to_string(k) + " could not prepare its share (its own call reports why)");
    // (before anything is sized by it: a share table is other ranks' — or a caller's — data)
    if (!geom_ok(g0) || g0.H > 65536 || g0.W > 65536) return fail(GPUART_HIP_ERR_ARG, "gather: inconsistent shares (frame of rank 0)");
    std::vector<uint8_t> cover(g0.H, 0);
    for (size_t k = 0; k < all.size(); k++) {
        const gpuart_tile_geom &g = all[k].g;
        if ((int)all[k].which != which || (int)all[k].root != root) return fail(GPUART_HIP_ERR_ARG, "gather: the ranks disagree about buffer or root");
        if (!geom_ok(g) || g.W != g0.W || g.H != g0.H) return fail(GPUART_HIP_ERR_ARG, "gather: inconsistent shares");
        if (g.th && (g.x0 != 0 || g.tw != g.W)) return fail(GPUART_HIP_ERR_ARG, "gather: shares must be full-width rows");
        for (uint32_t ly = 0; ly < g.th; ly++) {
            uint8_t &n = cover[gpuart_hip_frame_row(&g, ly)];
            if (n) return fail(GPUART_HIP_ERR_ARG, "gather: shares overlap (a frame row belongs to two ranks)");
            n = 1;
        }
    }
    for (uint32_t y = 0; y < g0.H; y++)
        if (!cover[y]) return fail(GPUART_HIP_ERR_ARG, "gather: frame row " + std::to_string(y) + " belongs to no rank");
    return 0;
}

/// What one rank's transfers need, by value: the closure that posts them may outlive the caller's frame (bounded.h).
struct GatherRank {
    int device, rank;
    ncclComm_t comm;
    hipStream_t stream;
    float4 *send, *stage;
    size_t tile_pixels;
};
GatherRank gather_rank(const gpuart_hip_ctx *c) { return GatherRank{c->device, c->comm_rank, c->comm, c->stream, c->d_send, c->d_stage, c->plan.tile_pixels}; }

/// One rank's transfers, between ncclGroupStart and ncclGroupEnd; nothing in here can fail for a reason of this rank alone
/// (buffers exist, shares are validated): a peer posts its send, the root its receives into the staging area.
ncclResult_t gather_post(const GatherRank &g, int root, const std::vector<GatherHello> &all) {
    const int n = (int)all.size(), me = g.rank;
    if (me != root) {
        if (!g.tile_pixels) return ncclSuccess;  // an empty share (more ranks than bands)
        return rccl()->Send(g.send, g.tile_pixels * 4, ncclFloat, root, g.comm, g.stream);
    }
    size_t off = 0;
    for (int k = 0; k < n; k++) {
        const size_t cnt = (size_t)all[k].g.tw * all[k].g.th;
        if (k != me && cnt) {
            const ncclResult_t e = rccl()->Recv(g.stage + off, cnt * 4, ncclFloat, k, g.comm, g.stream);
            if (e != ncclSuccess) return e;
        }
        off += cnt;
    }
    return ncclSuccess;
}

/// The group of transfers of the given ranks of this process (one for gpuart_hip_gather, all of them for _gather_all), posted
/// from ONE thread — RCCL's group state is per thread — under bounded(): ncclGroupEnd connects peers that have not talked before
/// and returns when that is done, i.e. when every peer has reached its own group; it has no timeout of its own.
int gather_transfers(const std::vector<GatherRank> &ranks, int root, const std::vector<GatherHello> &all) {
    const bounded_ns::Outcome o = bounded_ns::bounded("ncclGroupStart .. ncclGroupEnd (the gather's send / recv group)", bounded_ns::comm_timeout_ms(),
                                                      [ranks, root, all](std::string &d) {
        ncclResult_t e = rccl()->GroupStart();
        if (e != ncclSuccess) { d = std::string("ncclGroupStart: ") + rccl()->GetErrorString(e); return (int)e; }
        ncclResult_t pe = ncclSuccess;
        for (size_t k = 0; k < ranks.size() && pe == ncclSuccess; k++) {
            if (hipSetDevice(ranks[k].device) != hipSuccess) { pe = ncclUnhandledCudaError; break; }
            pe = gather_post(ranks[k], root, all);
        }
        const ncclResult_t ge = rccl()->GroupEnd();
        if (pe != ncclSuccess) { d = std::string("posting the transfers: ") + rccl()->GetErrorString(pe); return (int)pe; }
        if (ge != ncclSuccess) { d = std::string("ncclGroupEnd: ") + rccl()->GetErrorString(ge); return (int)ge; }
        return 0;
    });
    return comm_outcome(o, "gather");
}

/// After the group: the root scatters every share's rows into the full frame (its own straight from its send buffer).
int gather_place(gpuart_hip_ctx *c, const std::vector<GatherHello> &all, float4 *full) {
    size_t off = 0;
    for (size_t k = 0; k < all.size(); k++) {
        const gpuart_tile_geom &g = all[k].g;
        const size_t cnt = (size_t)g.tw * g.th;
        const float4 *src = (int)k == c->comm_rank ? c->d_send : c->d_stage + off;
        if (cnt) k_scatter_rows<<<dim3((unsigned)((cnt + 255) / 256)), 256, 0, c->stream>>>(g, src, full);
        off += cnt;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace
}  // extern "C++"

#ifdef GPUART_HIP_TEST_HOOKS
int gpuart_hip_test_share_table(const gpuart_tile_geom *shares, const uint32_t *status, int n, int which, int root) {
    if (!shares || n < 1 || n > 1024) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    try {
        std::vector<GatherHello> all((size_t)n);
        for (int k = 0; k < n; k++) {
            memset(&all[(size_t)k], 0, sizeof(GatherHello));
            all[(size_t)k].g = shares[k];
            all[(size_t)k].status = status ? status[k] : 0u;
            all[(size_t)k].which = (uint32_t)which; all[(size_t)k].root = (uint32_t)root;
        }
        return check_shares(all, which, root);
    } catch (const std::exception &e) {  // no exception crosses the C ABI
        return fail(GPUART_HIP_ERR_ARG, std::string("share table: ") + e.what());
    }
}
#endif

int gpuart_hip_comm_library(char *path, size_t size) {
    if (!path || size < 2) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    int r = need_rccl();
    if (r) return r;
    Dl_info di;
    memset(&di, 0, sizeof di);
    if (!dladdr((void *)rccl()->CommInitRank, &di) || !di.dli_fname) return fail(GPUART_HIP_ERR_DEVICE, "dladdr(ncclCommInitRank) failed");
    snprintf(path, size, "%s", di.dli_fname);
    return 0;
}

int gpuart_hip_comm_info(gpuart_hip_ctx *c, int *nranks, int *rank) {
    if (!c || !c->comm) return fail(GPUART_HIP_ERR_ARG, "no communicator (gpuart_hip_comm_init)");
    int r = need_rccl();
    if (r) return r;
    int n = 0, me = 0;
    NCCL_TRY(rccl()->CommCount(c->comm, &n));
    NCCL_TRY(rccl()->CommUserRank(c->comm, &me));
    if (nranks) *nranks = n;
    if (rank) *rank = me;
    return 0;
}

int gpuart_hip_wait(gpuart_hip_ctx *c, uint32_t timeout_ms) {
    if (!c) return fail(GPUART_HIP_ERR_ARG, "ctx == NULL");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = gpuart_hip_flush(c); if (fr) return fr; }
    // the pass lanes' work reaches the primary stream through their events (accumulation in pass order), so the primary stream
    // is the last to finish
    int r = wait_stream(c, timeout_ms, "gpuart_hip_wait");
    return r ? r : drain(c);
}

int gpuart_hip_gather(gpuart_hip_ctx *c, int which, float divide_by, int root, void *full_frame_device) {
    if (!c || !c->comm) return fail(GPUART_HIP_ERR_ARG, "no communicator (gpuart_hip_comm_init)");
    if (c->comm_nranks > 1024) return fail(GPUART_HIP_ERR_ARG, "more than 1024 ranks");
    HIP_TRY(hipSetDevice(c->device));
    const int n = c->comm_nranks;
    // 1. everything that can fail here alone; the verdict is part of what the ranks exchange
    GatherHello own;
    int mine = 0;
    if (root < 0 || root >= n || (which != 0 && which != 1)) mine = fail(GPUART_HIP_ERR_ARG, "bad argument");
    else if (c->comm_rank == root && !full_frame_device) mine = fail(GPUART_HIP_ERR_ARG, "the root needs a frame buffer");
    else mine = gather_prepare(c, which, divide_by, root, own);
    const std::string my_error = mine ? g_last_error : std::string();
    if (mine) { memset(&own, 0, sizeof own); own.which = (uint32_t)which; own.root = (uint32_t)root; }
    own.status = mine ? 1u : 0u;
    if (!c->d_hello && hipMalloc(&c->d_hello, (size_t)(1 + 1024) * sizeof(GatherHello)) != hipSuccess)
        return fail(GPUART_HIP_ERR_DEVICE, "gather: no device memory for the share table (the other ranks are left waiting: nothing can be told to them)");
    if (!c->h_hello && hipHostMalloc(&c->h_hello, (size_t)(1 + 1024) * sizeof(GatherHello), hipHostMallocDefault) != hipSuccess)
        return fail(GPUART_HIP_ERR_DEVICE, "gather: no pinned host memory for the share table (the other ranks are left waiting: nothing can be told to them)");
    // 2. everybody's share and status, through the communicator itself (12 words per rank). Both host ends of the two copies are
    //    the context's pinned table: the copies are queued, not performed, by these calls — the wait below is the only place this
    //    rank can block, and it is bounded —, and a call that gives up leaves no queued copy pointing at its own stack or heap.
    GatherHello *d = (GatherHello *)c->d_hello, *hh = (GatherHello *)c->h_hello;
    hh[0] = own;
    int r;
    if ((r = comm_layer_ok())) return r;
    HIP_TRY(hipMemcpyAsync(d, hh, sizeof own, hipMemcpyHostToDevice, c->stream));
    {   // (enqueueing a collective may itself wait — the first use of a channel —: under bounded(), bounded.h)
        const GatherRank me = gather_rank(c);
        const bounded_ns::Outcome o = bounded_ns::bounded("ncclAllGather (share exchange)", bounded_ns::comm_timeout_ms(), [me, d](std::string &dt) {
            (void)hipSetDevice(me.device);
            const ncclResult_t e = rccl()->AllGather(d, d + 1, sizeof(GatherHello) / 4, ncclUint32, me.comm, me.stream);
            if (e != ncclSuccess) dt = rccl()->GetErrorString(e);
            return (int)e;
        });
        if ((r = comm_outcome(o, "ncclAllGather"))) return r;
    }
    HIP_TRY(hipMemcpyAsync(hh + 1, d + 1, (size_t)n * sizeof own, hipMemcpyDeviceToHost, c->stream));
    if ((r = wait_stream(c, c->gather_timeout_ms, "gather: exchange of the shares"))) return r;
    if (mine) return fail(mine, my_error);
    std::vector<GatherHello> all(hh + 1, hh + 1 + n);
    // 3. the same table on every rank: the same verdict on every rank
    if ((r = check_shares(all, which, root))) return r;
    // 4. the transfers
    if ((r = gather_transfers({gather_rank(c)}, root, all))) return r;
    if (c->comm_rank == root) return gather_place(c, all, (float4 *)full_frame_device);
    return 0;
}

int gpuart_hip_gather_all(gpuart_hip_ctx *const *ctxs, int n, int which, float divide_by, int root, void *full_frame_device) {
    if (!ctxs || n < 1 || root < 0 || root >= n || !full_frame_device || (which != 0 && which != 1)) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    int r;
    std::vector<GatherHello> all((size_t)n);
    // one thread drives every rank: all local preparation first, transfers only when every rank is ready
    for (int k = 0; k < n; k++) {
        if (!ctxs[k] || !ctxs[k]->comm || ctxs[k]->comm_nranks != n || ctxs[k]->comm_rank != k || ctxs[k]->comm_group != ctxs[0]->comm_group)
            return fail(GPUART_HIP_ERR_NO_COMM, "contexts are not the ranks 0..n-1 of one communicator (gpuart_hip_comm_init_all)");
    }
    for (int k = 0; k < n; k++) {
        HIP_TRY(hipSetDevice(ctxs[k]->device));
        if ((r = gather_prepare(ctxs[k], which, divide_by, root, all[(size_t)k]))) return r;
    }
    if ((r = check_shares(all, which, root))) return r;
    if ((r = comm_layer_ok())) return r;
    std::vector<GatherRank> ranks;
    for (int k = 0; k < n; k++) ranks.push_back(gather_rank(ctxs[k]));
    if ((r = gather_transfers(ranks, root, all))) return r;
    HIP_TRY(hipSetDevice(ctxs[root]->device));
    return gather_place(ctxs[root], all, (float4 *)full_frame_device);
}

int gpuart_hip_gather_all_read(gpuart_hip_ctx *const *ctxs, int n, int which, float divide_by, int root, float *full_frame_host) {
    if (!ctxs || n < 1 || root < 0 || root >= n || !ctxs[root] || !full_frame_host) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    gpuart_hip_ctx *c = ctxs[root];
    const size_t bytes = (size_t)c->frame.W * c->frame.H * sizeof(float4);
    HIP_TRY(hipSetDevice(c->device));
    int r = ensure_scratch(c, bytes);
    if (r) return r;
    if ((r = gpuart_hip_gather_all(ctxs, n, which, divide_by, root, c->d_scratch))) return r;
    HIP_TRY(hipSetDevice(c->device));
    // The copy is queued behind the root's receives: waiting for it is waiting for every peer's rows. Bounded like the exchange of
    // gpuart_hip_gather (this wait was a plain hipStreamSynchronize until round 5 — one of the three unbounded waits on the path
    // of the gpuart_cli child that once did not end, profiles/r05/stall_path.txt). The device-to-host copy goes into PINNED memory
    // owned by the context: into the caller's pageable buffer hipMemcpyAsync is not asynchronous — the call itself would wait for
    // the stream, ahead of the bounded wait (the first version of this fix did exactly that; the test that holds the stream caught
    // it) — and a call that gives up leaves no queued copy pointing at memory the caller may free.
    if (c->h_frame_bytes < bytes) {
        // (hipHostFree waits for the device: if an earlier read-out of this context gave up, its copy into h_frame may still be queued
        // behind the stream that held it — that wait gets its bound first, and the buffer stays if the stream is still held)
        if (c->h_frame && c->abandoned && (r = wait_stream(c, c->gather_timeout_ms, "gather: an abandoned read-back still owns the pinned frame buffer"))) return r;
        if (c->h_frame) { (void)hipHostFree(c->h_frame); c->h_frame = nullptr; c->h_frame_bytes = 0; }
        if (hipHostMalloc(&c->h_frame, bytes, hipHostMallocDefault) != hipSuccess) { c->h_frame = nullptr; return fail(GPUART_HIP_ERR_DEVICE, "gather: no pinned host memory for the frame"); }
        c->h_frame_bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(c->h_frame, c->d_scratch, bytes, hipMemcpyDeviceToHost, c->stream));
    if ((r = wait_stream(c, c->gather_timeout_ms, "gather: the peers' rows and the read-back of the frame"))) return r;
    memcpy(full_frame_host, c->h_frame, bytes);
    return 0;
}

// ---- uploader hook: what gpuart_hip_upload_bvh decides about a tree, without a device (include/gpuart_hip.h) ----------
#ifdef GPUART_HIP_TEST_HOOKS
int gpuart_hip_test_tree_class(const float *quads, size_t nquads, uint32_t *flags) {
    if (!quads || !nquads || !flags) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    try {
        Converter cv;
        cv.q = quads; cv.nq = nquads;
        Converter::Child root;
        if (!cv.convert(root, 1)) return fail(GPUART_HIP_ERR_ARG, "malformed compiled BVH: " + cv.err);
        *flags = (cv.irregular ? 1u : 0u) | (cv.disorderly ? 2u : 0u) | (cv.subnormal ? 4u : 0u) | (cv.type_mask << 8);
        return 0;
    } catch (const std::exception &e) {
        return fail(GPUART_HIP_ERR_ARG, std::string("tree: ") + e.what());
    }
}
#endif

#ifdef GPUART_HIP_TEST_HOOKS
int gpuart_hip_test_tree_slack(const float *quads, size_t nquads, float *slack) {
    if (!quads || !nquads || !slack) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    try {
        Converter cv;
        cv.q = quads; cv.nq = nquads;
        Converter::Child root;
        if (!cv.convert(root, 1)) return fail(GPUART_HIP_ERR_ARG, "malformed compiled BVH: " + cv.err);
        *slack = tree_slack(cv, root);
        return 0;
    } catch (const std::exception &e) {
        return fail(GPUART_HIP_ERR_ARG, std::string("tree: ") + e.what());
    }
}
#endif

// ---- run planner hook: the planner of the context, driven without a device (include/gpuart_hip.h) ---------------------
#ifdef GPUART_HIP_TEST_HOOKS
int gpuart_hip_test_planner(const uint32_t cfg[8], const uint32_t *ops, int n_ops, uint32_t *runs, int max_runs) {
    if (!cfg || (!ops && n_ops) || n_ops < 0 || (!runs && max_runs) || max_runs < 0) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    RunPlanner p;
    p.batch_limit = MAX_BATCH;
    if (cfg[0]) p.batch_limit = std::min<uint32_t>(cfg[0], MAX_BATCH);
    if (cfg[1]) p.lanes_total = std::min<uint32_t>(cfg[1], 32);
    if (cfg[2]) p.batch_paths = (size_t)cfg[2] << 20;
    if (cfg[3]) p.min_run_paths = (size_t)cfg[3] << 10;
    if (cfg[4]) p.small_paths = cfg[4] == 0xffffffffu ? 0 : (size_t)cfg[4] << 10;
    if (cfg[5]) p.lane_budget = (size_t)cfg[5] << 20;
    if (cfg[6]) p.plan_run_factor = cfg[6] / 100.0;
    uint32_t W = 0, H = 0, alloc_fails = 0;
    int n_runs = 0;
    auto record = [&](int op, size_t count) {
        if (n_runs < max_runs) {
            uint32_t *o = runs + 6 * (size_t)n_runs;
            o[0] = (uint32_t)op; o[1] = p.n_slots; o[2] = p.max_batch; o[3] = (uint32_t)count;
            o[4] = p.uses_run_kernel(count) ? 1u : 0u; o[5] = (uint32_t)p.pending;
        }
        n_runs++;
    };
    auto flush = [&](int op) { for (size_t count : p.on_flush()) record(op, count); };
    auto tile = [&](int op, uint32_t tw, uint32_t th) -> int {
        flush(op);  // realloc_tile: what is pending belongs to the old tile
        const size_t n = (size_t)((tw + 7) / 8) * ((th + 7) / 8) * 64;
        if (n > 0xfffffff0ull) return GPUART_HIP_ERR_ARG;
        p.set_tile((uint32_t)n, (size_t)tw * th);
        for (; alloc_fails; alloc_fails--)
            if (!p.shrink()) return GPUART_HIP_ERR_DEVICE;
        return 0;
    };
    for (int k = 0; k < n_ops; k++) {
        const uint32_t op = ops[3 * k], a = ops[3 * k + 1], b = ops[3 * k + 2];
        int r = 0;
        switch (op) {
        case 0:
            if (!a || !b || a > 65536 || b > 65536) return fail(GPUART_HIP_ERR_ARG, "bad frame size");
            W = a; H = b;
            r = tile(k, W, H);
            break;
        case 1: {
            gpuart_tile_geom g;
            if (!W || gpuart_hip_share_of_rank(W, H, (int)a, (int)b, 8, &g) || !g.th) return fail(GPUART_HIP_ERR_ARG, "bad share");
            r = tile(k, g.tw, g.th);
            break;
        }
        case 2: p.set_plan(a); break;
        case 3:
            if (a > 5) return fail(GPUART_HIP_ERR_ARG, "bad mode");
            flush(k); p.mode = (int)a; p.plan();
            break;
        case 4:
            if (!p.n_slots) return fail(GPUART_HIP_ERR_ARG, "no frame size set");
            for (uint32_t i = 0; i < a; i++) {
                if (p.mode == 2) continue;  // the megakernel renders a pass at once: nothing is collected
                if (const size_t count = p.on_pass()) record(k, count);
            }
            break;
        case 5: flush(k); break;
        case 6: alloc_fails = a; break;
        default: return fail(GPUART_HIP_ERR_ARG, "unknown planner op");
        }
        if (r) return fail(r, "planner: tile refused");
    }
    return n_runs;
}
#endif

// ---- test hooks ------------------------------------------------------------------------------------
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), 256, 0, c->stream

#ifdef GPUART_HIP_TEST_HOOKS
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
    return run_hook(c, n, ins, 4, outs, 1, [&](auto &i, auto &o) { k_test_aabb<<<GRID1(n)>>>(i[0], i[1], i[2], i[3], n, o[0], (int)c->quick_boxes); });
}
int gpuart_hip_test_traverse(gpuart_hip_ctx *c, const float *rs, const float *rd, const float us[4], int n, int any_hit,
                             float *out0, float *out1) {
    if (!c || !c->have_scene || !us) return fail(GPUART_HIP_ERR_ARG, "no scene uploaded");
    if (any_hit == 2 && c->exact_boxes) return fail(GPUART_HIP_ERR_ARG, "a tree with irregular boxes (or boxes that do not bound their contents) is only walked in the reference's order");
    Float4Arg a; memcpy(a.v, us, 16);
    Scene sc = scene_of(c);
    const float *ins[] = {rs, rd}; float *outs[] = {out0, out1};
    return run_hook(c, n, ins, 2, outs, 2, [&](auto &i, auto &o) {
        dim3 grid(std::min<uint32_t>(c->grid_waves, (uint32_t)((n + BLOCK - 1) / BLOCK)));
        if (any_hit == 2) k_test_traverse<false, true><<<grid, BLOCK, 0, c->stream>>>(sc, i[0], i[1], a, n, o[0], o[1], c->d_spill);
        else if (any_hit) k_test_traverse<true><<<grid, BLOCK, 0, c->stream>>>(sc, i[0], i[1], a, n, o[0], o[1], c->d_spill);
        else k_test_traverse<false><<<grid, BLOCK, 0, c->stream>>>(sc, i[0], i[1], a, n, o[0], o[1], c->d_spill);
    });
}
int gpuart_hip_test_cam_rays(gpuart_hip_ctx *c, float *rstart, float *rdir) {
    if (!c || !c->have_camera || !c->plan.tile_pixels) return fail(GPUART_HIP_ERR_ARG, "camera / frame not set");
    int n = (int)c->plan.tile_pixels;
    float *outs[] = {rstart, rdir};
    Frame f = c->frame;
    return run_hook(c, n, nullptr, 0, outs, 2, [&](auto &, auto &o) { k_test_cam_rays<<<GRID1(n)>>>(f, o[0], o[1]); });
}
#endif

namespace {
/// One wave that keeps the stream busy for `ticks` of the 100 MHz wall clock (gpuart_hip_test_stall): it ends by itself.
__global__ void k_test_stall(unsigned long long ticks, unsigned long long *sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long n = 0;
    while (wall_clock64() - t0 < ticks) n++;
    if (sink && threadIdx.x == 0) *sink = n;
}
}  // namespace
#ifdef GPUART_HIP_TEST_HOOKS
int gpuart_hip_test_tile_order(gpuart_hip_ctx *c, const uint32_t *order, size_t n) {
    if (!c || !c->frame.W) return fail(GPUART_HIP_ERR_ARG, "no frame size set");
    HIP_TRY(hipSetDevice(c->device));
    int r = gpuart_hip_flush(c);
    if (r || (r = drain(c))) return r;
    if (!order) { c->frame.tile_order = nullptr; c->order_from_hook = false; c->order_cur = -1; c->order_sorting = false; c->order_stale = true; return 0; }
    const size_t tiles = (size_t)((c->frame.tw + 7) / 8) * ((c->frame.th + 7) / 8);
    if (n != tiles) return fail(GPUART_HIP_ERR_ARG, "the order must name every 8x8 block of the tile once");
    std::vector<bool> seen(tiles, false);
    for (size_t k = 0; k < n; k++) {
        if (order[k] >= tiles || seen[order[k]]) return fail(GPUART_HIP_ERR_ARG, "the order must name every 8x8 block of the tile once");
        seen[order[k]] = true;
    }
    if (!c->d_tile_order[0]) HIP_TRY(hipMalloc(&c->d_tile_order[0], tiles * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(c->d_tile_order[0], order, tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->frame.tile_order = c->d_tile_order[0];
    c->order_from_hook = true; c->order_sorting = false;
    return 0;
}
int gpuart_hip_test_current_tile_order(gpuart_hip_ctx *c, uint32_t *order, size_t n) {
    if (!c || !c->frame.W) return fail(GPUART_HIP_ERR_ARG, "no frame size set");
    HIP_TRY(hipSetDevice(c->device));
    int r = gpuart_hip_flush(c);
    if (r || (r = drain(c))) return r;
    if (c->order_sorting) {  // drained: the sort has finished
        c->order_cur = c->order_next;
        c->order_sorting = false;
    }
    const uint32_t *cur = c->order_from_hook ? c->frame.tile_order : c->order_cur >= 0 ? c->d_tile_order[c->order_cur] : nullptr;
    if (!cur) return 0;
    const size_t tiles = (size_t)((c->frame.tw + 7) / 8) * ((c->frame.th + 7) / 8);
    if (!order || n != tiles) return fail(GPUART_HIP_ERR_ARG, "room for one entry per 8x8 block of the tile, please");
    HIP_TRY(hipMemcpy(order, cur, tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 1;
}
int gpuart_hip_test_sort_tiles(gpuart_hip_ctx *c, const uint32_t *cost, size_t n, uint32_t *order) {
    if (!c || !cost || !order || n == 0 || n > 0x04000000u) return fail(GPUART_HIP_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc(&d, 2 * n * sizeof(uint32_t)));
    hipError_t e = hipMemcpyAsync(d, cost, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        k_tile_order<<<1, TO_THREADS, 0, c->stream>>>(d, d + n, (uint32_t)n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(order, d + n, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIP_TRY(e);
    return 0;
}
#endif
// ---- phase watchdog and the bounded-call mechanism (bounded.h; pure host code) ----------------------------------------------
int gpuart_hip_phase_begin(const char *name, uint32_t timeout_ms) { return bounded_ns::phase_begin(name, timeout_ms); }
int gpuart_hip_phase_end(void) { return bounded_ns::phase_end(); }
int gpuart_hip_phase_log(int on) { return bounded_ns::phase_log(on); }
int gpuart_hip_comm_stuck(void) { return bounded_ns::stuck().load() ? 1 : 0; }
#ifdef GPUART_HIP_TEST_HOOKS
int gpuart_hip_test_bounded_call(uint32_t hold_ms, uint32_t timeout_ms, int mark_stuck) {
    // (the helper's own `stuck` mark is global: the hook restores it unless the test wants to see the layer refuse its entry points)
    const bool was = bounded_ns::stuck().load();
    const bounded_ns::Outcome o = bounded_ns::bounded("gpuart_hip_test_bounded_call", timeout_ms, [hold_ms](std::string &d) {
        std::this_thread::sleep_for(std::chrono::milliseconds(hold_ms));
        d = "held for " + std::to_string(hold_ms) + " ms";
        return 0;
    });
    if (!mark_stuck && !was) bounded_ns::stuck().store(false);
    if (o.timed_out) return fail(GPUART_HIP_ERR_TIMEOUT, o.detail);
    return o.rc;
}
#endif

#ifdef GPUART_HIP_TEST_HOOKS
int gpuart_hip_test_stall(gpuart_hip_ctx *c, uint32_t ms) {
    if (!c || ms > 5000) return fail(GPUART_HIP_ERR_ARG, "bad argument (at most 5000 ms)");
    HIP_TRY(hipSetDevice(c->device));
    k_test_stall<<<1, 64, 0, c->stream>>>((unsigned long long)ms * 100000ull, nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}
#endif

#ifdef GD_STEP_STATS
/// diagnostic builds only: reads (and clears) k_trace's step statistics (kernels_pipeline.h): box steps, lanes in them, leaf steps, lanes in
/// them, rounds of the wide loop, lanes holding a ray in them, refill episodes
int gpuart_hip_debug_step_stats(gpuart_hip_ctx *c, unsigned long long *out) {
    if (!c || !out) return GPUART_HIP_ERR_ARG;
    static unsigned long long zero[8];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_stats), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_step_stats), zero, sizeof(zero)));
    HIP_TRY(hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(gd::g_pop_stats), 4 * sizeof(unsigned long long)));   // out: 16 words
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gd::g_pop_stats), zero, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyFromSymbol(out + 12, HIP_SYMBOL(g_phase_ticks), 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), zero, 4 * sizeof(unsigned long long)));
    return 0;
}
#endif
#ifdef GD_QUICK_CHECK
/// diagnostic builds only: reads (and clears) the counters of the quick box answers checked against the six face tests (device_scene.h)
int gpuart_hip_debug_quick_stats(gpuart_hip_ctx *c, unsigned long long *out) {
    if (!c || !out) return GPUART_HIP_ERR_ARG;
    static unsigned long long zero[8];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(gd::g_quick_stats), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gd::g_quick_stats), zero, sizeof(zero)));
    return 0;
}
#endif
#ifdef GD_RUN_TIMELINE
/// diagnostic builds only: the per-wave timeline of the last k_run launch (kernel_run.h), 24 words per wave
int gpuart_hip_debug_run_timeline(gpuart_hip_ctx *c, unsigned long long *out, size_t waves) {
    if (!c || !out || waves > 8192) return GPUART_HIP_ERR_ARG;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_run_timeline), waves * 24 * sizeof(unsigned long long)));
    return 0;
}
/// diagnostic builds only: reads (and clears) the traversal-stack event counters (pushes, spilling pushes, pops, reloading pops)
int gpuart_hip_debug_stack_events(gpuart_hip_ctx *c, unsigned long long *out) {
    if (!c || !out) return GPUART_HIP_ERR_ARG;
    static unsigned long long zero[4];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(gd::g_stack_events), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gd::g_stack_events), zero, sizeof(zero)));
    return 0;
}
/// diagnostic builds only: reads (and clears) the busy-lane histogram of the k_run launches since the last call, 256 words
int gpuart_hip_debug_run_hist(gpuart_hip_ctx *c, unsigned long long *out) {
    if (!c || !out) return GPUART_HIP_ERR_ARG;
    static unsigned long long zero[256];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_run_hist), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_run_hist), zero, sizeof(zero)));
    return 0;
}
#endif
}  // extern "C"
