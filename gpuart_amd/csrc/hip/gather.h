// gather.h — screen-space shares of one frame and the multi-GPU gather of the rendered rows,
// part of libgpuart_hip.so (included by gpuart_hip.hip).
//
// A frame is sharded by rows: bands of `band_rows` rows dealt round-robin to the ranks (gpuart_hip_share_of_rank), nothing
// is exchanged per pass, and once the passes are done every rank sends its normalised rows to one root over RCCL
// (point-to-point: on an 8-GPU MI355X node every peer owns a direct xGMI link to the root), where a small kernel scatters
// them into the full frame. This is the step the reference performs with its ptracingNormalize draw to the default
// framebuffer (src/renderer.cpp:601-616, shaders/pt_normalize.glsl:44-47), for a frame that lives on several GPUs.
//
// RCCL is loaded with dlopen the first time a communicator is made, so the library (and every single-GPU caller) does not
// depend on it. The layout helpers are plain host functions: the CPU tests drive them over gloo with world size 2.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "gpuart_hip.h"

namespace {

// ---- RCCL entry points, resolved at run time -------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // GPUART_HIP_RCCL_LIBRARY: this file and no other (a site's own build of RCCL; the tests' stand-in whose calls can be made
        // to never return and which serves N ranks inside one process — tests/stubs/rccl_stub.cpp — which is how the bounded waits and the
        // N > 1 transfers are exercised on one GPU)
        if (const char *own = getenv("GPUART_HIP_RCCL_LIBRARY")) {
            r.lib = dlopen(own, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) { r.err = std::string("cannot load GPUART_HIP_RCCL_LIBRARY: ") + dlerror(); return; }
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) { r.err = std::string("cannot load librccl: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

/// Rows of one share, placed into the full frame: full[frame_row(ly)][x0 + lx] = tile[ly][lx].
__global__ void k_scatter_rows(gpuart_tile_geom g, const float4 *__restrict__ tile, float4 *__restrict__ full) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)g.tw * g.th;
    if (i >= n) return;
    const uint32_t ly = (uint32_t)(i / g.tw), lx = (uint32_t)(i % g.tw);
    const uint32_t fy = g.y0 + (ly / g.band_rows) * g.band_stride + ly % g.band_rows;
    full[(size_t)fy * g.W + g.x0 + lx] = tile[i];
}

bool geom_ok(const gpuart_tile_geom &g) {
    if (!g.W || !g.H || !g.tw || !g.band_rows || g.band_stride < g.band_rows || (uint64_t)g.x0 + g.tw > g.W) return false;
    if (!g.th) return true;  // an empty share (more ranks than bands)
    const uint64_t last = (uint64_t)g.y0 + (uint64_t)((g.th - 1) / g.band_rows) * g.band_stride + (g.th - 1) % g.band_rows;
    return last < g.H;
}

}  // namespace

extern "C" {

// ---- layout helpers: pure host code --------------------------------------------------------------------------------------
int gpuart_hip_share_of_rank(uint32_t W, uint32_t H, int rank, int nranks, uint32_t band_rows, gpuart_tile_geom *out) {
    if (!out || !W || !H || nranks < 1 || rank < 0 || rank >= nranks || !band_rows) return GPUART_HIP_ERR_ARG;
    const uint32_t nbands = (H + band_rows - 1) / band_rows;
    uint32_t rows = 0;
    for (uint32_t bnd = (uint32_t)rank; bnd < nbands; bnd += (uint32_t)nranks)
        rows += std::min(band_rows, H - bnd * band_rows);
    out->W = W; out->H = H; out->x0 = 0; out->tw = W;
    out->y0 = (uint32_t)rank * band_rows;
    out->th = rows;
    out->band_rows = band_rows;
    out->band_stride = band_rows * (uint32_t)nranks;
    return 0;
}

uint32_t gpuart_hip_frame_row(const gpuart_tile_geom *g, uint32_t local_row) {
    return g->y0 + (local_row / g->band_rows) * g->band_stride + local_row % g->band_rows;
}

int gpuart_hip_scatter_rows_host(const gpuart_tile_geom *g, const float *tile_rgba, float *full_rgba) {
    if (!g || !tile_rgba || !full_rgba || !geom_ok(*g)) return GPUART_HIP_ERR_ARG;
    for (uint32_t ly = 0; ly < g->th; ly++)
        memcpy(full_rgba + ((size_t)gpuart_hip_frame_row(g, ly) * g->W + g->x0) * 4, tile_rgba + (size_t)ly * g->tw * 4,
               (size_t)g->tw * 4 * sizeof(float));
    return 0;
}

}  // extern "C"
