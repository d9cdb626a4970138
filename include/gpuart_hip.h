/* gpuart_hip.h — C ABI of libgpuart_hip.so, the MI355X (gfx950) device back end of gpuart.
 *
 * Drop-in boundary: this library replaces the reference's OpenGL wrapper layer
 * (reference src/gl_utils.h:108-379 — GL::Buffer/Texture/Shader/Program/Framebuffer and
 * GL::Utils::DrawFullscreenQuad) together with the GLSL programs the reference's Renderer
 * launches through it (reference src/renderer.cpp:259-359). The C++ `gpuart::Renderer`
 * (gpuart_amd/csrc/host/renderer.{h,cpp}) keeps the reference's public API and calls only the
 * functions below; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes, no C++ types, no exceptions across the boundary.
 * Every function returns 0 on success or a negative gpuart_hip_status; the message of the last
 * failure (per thread) is available from gpuart_hip_last_error(). One context per device;
 * contexts are independent. Host buffers are caller-owned. Work is asynchronous on the
 * context's own HIP stream until gpuart_hip_finish() / a read-back.
 *
 * Image convention (as the reference's full-screen quad): row 0 is the BOTTOM row; pixels are
 * RGBA32F; a context renders only its tile [x0,x0+tw) x [y0,y0+th) of the W x H frame and its
 * buffers are tile-sized.
 */
#ifndef GPUART_HIP_H
#define GPUART_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpuart_hip_ctx gpuart_hip_ctx;

typedef enum gpuart_hip_status {
    GPUART_HIP_OK = 0,
    GPUART_HIP_ERR_ARG = -1,      /* bad argument / call order (e.g. render before upload) */
    GPUART_HIP_ERR_DEVICE = -2,   /* a HIP runtime call failed */
    GPUART_HIP_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    GPUART_HIP_ERR_TIMEOUT = -4,  /* a bounded wait ran out (gpuart_hip_wait, the waits of the gather, an RCCL call that did not return): a peer is missing */
    GPUART_HIP_ERR_NO_COMM = -5   /* the contexts are not (or no longer) the ranks of one communicator: make one and call again */
} gpuart_hip_status;

/* The uniforms of the reference's directLighting / pathTracing programs
 * (reference src/renderer.cpp:51-79,383-398,567-589; shaders/path_tracing.glsl:82-108). */
typedef struct gpuart_params {
    float sunDirAlt[4];       /* SunDirAlt: unit direction towards the Sun, altitude (rad) */
    int32_t sunEnabled;       /* SunDirectLightingEnabled */
    float userSphere[4];      /* UserSphere: centre, radius (radius 0 = disabled) */
    float userSphereEm[3];    /* UserSphereEm */
    uint32_t userSphereFlags; /* UserSphereFlags: 1 = EM_NONZERO, 2 = SPECULAR, 4 = FUZZY */
    float pixelSize;          /* PixelSize */
    float cameraPos[3];       /* CameraPos */
    int32_t maxSegments;      /* MAX_PATH_SEGMENTS, a shader const (5) in the reference; at most GPUART_HIP_MAX_SEGMENTS */
    float minWeight;          /* MIN_WEIGHT, a shader const (0.01) in the reference */
} gpuart_params;

#define GPUART_HIP_MAX_SEGMENTS 1024 /* gpuart_hip_pt_pass rejects larger maxSegments (one launch + counters per segment) */

/* Exact work counters (closest-hit queries as the reference performs them). */
typedef struct gpuart_counters {
    uint64_t rays;          /* closest-hit BVH queries: camera, bounce and shadow rays */
    uint64_t nodes;         /* distinct BVH nodes whose box was tested, summed over rays */
    uint64_t prim_tests[4]; /* tested primitives by type: sphere, disc, triangle, cone */
    uint64_t segments;      /* path segments traced */
    uint64_t box_steps;     /* interior-node visits (one 64-byte record fetch each) */
    uint64_t box_steps_top; /* ... of nodes above level GPUART_HIP_TOP_DEPTH (environment, read at upload; diagnostic) */
    uint64_t rewalks;       /* mode 4: closest-hit queries whose nearest-child-first walk could not certify its answer and that
                             * were walked again in the reference's order (their second walk's nodes / primitives are counted) */
} gpuart_counters;

const char *gpuart_hip_last_error(void);

/* Replaces GL::Init + the Renderer constructor's shader/program creation
 * (reference src/gl_utils.cpp:133-155, src/renderer.cpp:217-359). */
int gpuart_hip_create(int device, gpuart_hip_ctx **out);
int gpuart_hip_destroy(gpuart_hip_ctx *ctx);

/* Replaces Renderer::InitPerPixelTextures (reference src/renderer.cpp:169-194): frame size and
 * the tile this context owns (default: the whole frame). Clears the accumulator. */
int gpuart_hip_resize(gpuart_hip_ctx *ctx, uint32_t width, uint32_t height);
int gpuart_hip_set_tile(gpuart_hip_ctx *ctx, uint32_t x0, uint32_t y0, uint32_t tw, uint32_t th);
/* Row-interleaved tile for load-balanced multi-GPU sharding: the context owns th_local rows, local row ly being
 * frame row y0 + (ly / band_rows) * band_stride + ly % band_rows (e.g. rank r of N: y0 = 8r, band_rows = 8,
 * band_stride = 8N). Buffers and read-backs are tw x th_local, local row order. */
int gpuart_hip_set_tile_interleaved(gpuart_hip_ctx *ctx, uint32_t x0, uint32_t y0, uint32_t tw, uint32_t th_local,
                                    uint32_t band_rows, uint32_t band_stride);

/* Replaces the GL_TEXTURE_BUFFER upload of Renderer::SetPrimitives (reference
 * src/renderer.cpp:472-475). `quads` is the reference's canonical compiled tree
 * (BoundingVolumesHierarchy::Compile, reference src/bvh.cpp:161-222), nquads RGBA32F quads;
 * it is re-laid out on upload into the flat SoA device arrays described in DESIGN.md. */
int gpuart_hip_upload_bvh(gpuart_hip_ctx *ctx, const float *quads, size_t nquads);

/* Replaces the cameraInit program (reference shaders/cam_init.glsl:45-50, launched from
 * Renderer::SetCamera, src/renderer.cpp:151-161): stores Pos/BottomLeft/DeltaHorz/DeltaVert;
 * primary rays are regenerated in-kernel instead of being written to two ray textures. */
int gpuart_hip_set_camera(gpuart_hip_ctx *ctx, const float pos[3], const float bottomLeft[3],
                          const float deltaHorz[3], const float deltaVert[3]);

/* Replaces the directLighting draw (reference src/renderer.cpp:372-404,
 * shaders/direct_lighting.glsl:134-207). Result in the context's frame buffer. */
int gpuart_hip_render_direct(gpuart_hip_ctx *ctx, const gpuart_params *p);

/* Replaces the accumulator clear of Renderer::ResetPathTracing (reference src/renderer.cpp:490-500). */
int gpuart_hip_pt_reset(gpuart_hip_ctx *ctx);

/* Scheduling hint, no reference counterpart: how many gpuart_hip_pt_pass calls the caller expects to make before it
 * observes the result (Renderer::RestartPathTracing knows it: pathsPerPixel / pathsPerPass, reference
 * src/renderer.cpp:502-508). Sizes the pipeline runs the passes are grouped into; 0 = unknown. Never changes results. */
int gpuart_hip_pt_plan(gpuart_hip_ctx *ctx, uint32_t passes);

/* Replaces one pathTracing draw (reference src/renderer.cpp:534-599,
 * shaders/path_tracing.glsl:133-256): accum += radiance of `npaths` paths per pixel. */
int gpuart_hip_pt_pass(gpuart_hip_ctx *ctx, const gpuart_params *p, const float randSeed[4], int npaths);

/* Read-back (the reference draws to the GL framebuffer instead; ptracingNormalize,
 * reference shaders/pt_normalize.glsl:44-47, is the `divide_by` of the accumulator).
 * which: 0 = direct-lighting frame, 1 = path-tracing accumulator. divide_by <= 0 means 1.
 * rgba_host receives tw*th*4 floats. Synchronises the stream. */
int gpuart_hip_read(gpuart_hip_ctx *ctx, int which, float *rgba_host, float divide_by);

/* The inverse of gpuart_hip_read for which = 1: replaces the path-tracing accumulator of the tile with tw*th*4 floats
 * from the host (resuming a checkpointed progressive render; the reference has no equivalent). */
int gpuart_hip_write(gpuart_hip_ctx *ctx, int which, const float *rgba_host);

/* Same, device to device, into caller-owned device memory (for the multi-GPU gather: the caller
 * hands it to RCCL). Asynchronous on the context's stream; call gpuart_hip_finish before use. */
int gpuart_hip_export(gpuart_hip_ctx *ctx, int which, void *rgba_device, float divide_by);

/* ---- one frame on several GPUs ---------------------------------------------------------------
 * The frame is sharded by rows: bands of `band_rows` rows dealt round-robin to the ranks; ranks exchange nothing per pass;
 * when the passes are done every rank sends its rows to one root over RCCL (point-to-point over xGMI) and the root scatters
 * them into the full frame on its device. Replaces the reference's normalise-to-display step for a frame that lives on
 * several GPUs (reference src/renderer.cpp:601-616, shaders/pt_normalize.glsl:44-47: `divide_by`). */
typedef struct gpuart_tile_geom {
    uint32_t W, H;                    /* the frame */
    uint32_t x0, y0, tw, th;          /* the share: tw x th local pixels; th may be 0 (more ranks than bands) */
    uint32_t band_rows, band_stride;  /* local row ly = frame row y0 + (ly / band_rows) * band_stride + ly % band_rows */
} gpuart_tile_geom;

/* Pure host helpers (no device needed): rank's share of a W x H frame split over nranks in bands of band_rows rows (8 keeps
 * the 8x8 pixel tiles of the path state intact); frame row of a local row; host version of the root's scatter. */
int gpuart_hip_share_of_rank(uint32_t W, uint32_t H, int rank, int nranks, uint32_t band_rows, gpuart_tile_geom *out);
uint32_t gpuart_hip_frame_row(const gpuart_tile_geom *g, uint32_t local_row);
int gpuart_hip_scatter_rows_host(const gpuart_tile_geom *g, const float *tile_rgba, float *full_rgba);

/* The context renders share `g` of its frame (= gpuart_hip_set_tile_interleaved); the share it currently renders. */
int gpuart_hip_set_share(gpuart_hip_ctx *ctx, const gpuart_tile_geom *g);
int gpuart_hip_get_share(gpuart_hip_ctx *ctx, gpuart_tile_geom *g);

/* Communicator. One process per GPU: rank 0 obtains an id (gpuart_hip_comm_unique_id = ncclGetUniqueId), hands its
 * GPUART_HIP_UNIQUE_ID_BYTES bytes to the other ranks by any means (bench.py: torch.distributed broadcast), every rank calls
 * gpuart_hip_comm_init (= ncclCommInitRank on the context's device). One process driving several GPUs (gpuart_cli --gpus N):
 * gpuart_hip_comm_init_all over its contexts (= ncclCommInitAll; context k becomes rank k). A caller that already owns an
 * ncclComm_t attaches it instead (not destroyed with the context). RCCL is loaded on first use. */
#define GPUART_HIP_UNIQUE_ID_BYTES 128
int gpuart_hip_comm_unique_id(void *id128);
int gpuart_hip_comm_init(gpuart_hip_ctx *ctx, int nranks, int rank, const void *id128);
int gpuart_hip_comm_attach(gpuart_hip_ctx *ctx, void *nccl_comm, int nranks, int rank);
int gpuart_hip_comm_init_all(gpuart_hip_ctx *const *ctxs, int n);
int gpuart_hip_comm_destroy(gpuart_hip_ctx *ctx);
/* Which RCCL the library resolved its entry points from (dladdr of ncclCommInitRank): a process that also holds another
 * framework's RCCL — torch.distributed's, say — can record that both are the same file. Loads RCCL if nothing has yet. */
int gpuart_hip_comm_library(char *path, size_t size);
/* What the communicator itself says (ncclCommCount, ncclCommUserRank); GPUART_HIP_ERR_ARG without one. */
int gpuart_hip_comm_info(gpuart_hip_ctx *ctx, int *nranks, int *rank);

/* Collective over the communicator: every rank contributes buffer `which` (0 direct lighting, 1 accumulator) of its share,
 * divided by `divide_by`; on `root`, full_frame_device (W*H*4 floats in the root's device memory, frame row order) receives
 * the assembled frame (other ranks pass NULL). Shares are exchanged through the communicator itself. Asynchronous on the
 * contexts' streams after a short host synchronisation: call gpuart_hip_finish on the root before using the frame.
 * gpuart_hip_gather_all is the same for the ranks 0..n-1 of a gpuart_hip_comm_init_all communicator, from one thread. */
/* Failure behaviour: everything that can fail on one rank alone (pending passes, buffers, arguments) happens BEFORE the
 * ranks commit to a transfer, and its verdict travels with the share, so a rank that cannot take part makes every rank
 * return an error rather than leaving its peers blocked in a receive. The shares must be full-width rows covering every frame
 * row exactly once (gpuart_hip_share_of_rank; a rank beyond the number of bands holds an empty share and sends nothing). The
 * host-side wait for the share exchange is bounded (GPUART_HIP_GATHER_TIMEOUT_MS, default 60 000, 0 = unbounded):
 * GPUART_HIP_ERR_TIMEOUT means a peer never entered the collective.
 * EVERY host-side wait of this section is bounded (round 5; until then only the share exchange of gpuart_hip_gather was):
 *   - waits on the context's stream — the share exchange, the read-back of gpuart_hip_gather_all_read, gpuart_hip_comm_destroy
 *     and gpuart_hip_destroy of a context whose gather gave up — by GPUART_HIP_GATHER_TIMEOUT_MS;
 *   - RCCL calls that return only when peers or RCCL's bootstrap play along — ncclCommInitRank / ncclCommInitAll, the
 *     ncclAllGather enqueue, ncclGroupStart..ncclGroupEnd of the transfers, ncclCommDestroy — by GPUART_HIP_COMM_TIMEOUT_MS
 *     (default 120 000, 0 = unbounded): they run on a helper thread; when the bound runs out the call returns
 *     GPUART_HIP_ERR_TIMEOUT naming the RCCL call, the helper stays parked in it, and from then on every entry point of this
 *     section fails at once with GPUART_HIP_ERR_TIMEOUT (gpuart_hip_comm_stuck() == 1): end the process with _exit.
 * GPUART_HIP_RCCL_LIBRARY names the RCCL to load instead of the system's librccl.so.1. */
int gpuart_hip_gather(gpuart_hip_ctx *ctx, int which, float divide_by, int root, void *full_frame_device);
int gpuart_hip_gather_all(gpuart_hip_ctx *const *ctxs, int n, int which, float divide_by, int root, void *full_frame_device);
/* gpuart_hip_gather_all into host memory (W*H*4 floats): the frame is assembled on the root's device, then read back. */
int gpuart_hip_gather_all_read(gpuart_hip_ctx *const *ctxs, int n, int which, float divide_by, int root, float *full_frame_host);

/* 1 once an RCCL call of this process has outlived GPUART_HIP_COMM_TIMEOUT_MS (see above), else 0. */
int gpuart_hip_comm_stuck(void);

/* Phase watchdog (pure host code; no reference counterpart — the reference's render loop, src/main.cpp:549-599, has nothing
 * that waits for another process). The caller names what it is about to do and how long that may take; gpuart_hip_phase_end
 * disarms. A watcher thread that sees the phase outlive its bound writes the phase, how long it has run and the most recent
 * error messages of libgpuart_hip.so of ALL threads to stderr and ends the process with _exit(86) — no unwinding, nothing is
 * re-executed. One phase at a time (a second begin replaces the first). timeout_ms 0: only the phase lines. Every begin / end
 * writes one line to stderr unless GPUART_HIP_PHASE_LOG=0. gpuart_cli --gpus N and the ranks of bench.py bracket communicator
 * creation, the gather and communicator destruction with it. */
int gpuart_hip_phase_begin(const char *name, uint32_t timeout_ms);
int gpuart_hip_phase_end(void);
/* Phase lines on (1) / off (0) from now on, overriding GPUART_HIP_PHASE_LOG; returns the previous setting. The watchdog is not
 * affected: a caller that brackets something inside a timed region (bench.py's gathers) keeps the bound and drops the two lines. */
int gpuart_hip_phase_log(int on);
/* glFlush() equivalent. gpuart_hip_pt_pass may only *collect* a pass: passes with identical parameters are launched
 * together as one run of the pipeline (path slot = pixel x passes of the run + pass, up to 16M paths, so that the persistent
 * BVH-query waves take many rays per lane and same-pixel rays of different passes share a wavefront), and up to 8 runs are in flight at once on separate HIP streams (fewer when their path state would
 * exceed 16 GB); results are accumulated strictly in pass order. Everything
 * that observes or changes state (read, export, finish, reset, set_camera, ...) flushes by itself. */
int gpuart_hip_flush(gpuart_hip_ctx *ctx);

/* glFinish() equivalent (reference src/main.cpp:564,584). */
int gpuart_hip_finish(gpuart_hip_ctx *ctx);
/* gpuart_hip_finish with a bound: GPUART_HIP_ERR_TIMEOUT if the context's work (e.g. the transfers of a gather whose peer
 * died) is not complete after timeout_ms milliseconds (0: no bound). Nothing is cancelled: the caller decides (bench.py exits). */
int gpuart_hip_wait(gpuart_hip_ctx *ctx, uint32_t timeout_ms);

/* Execution mode of gpuart_hip_pt_pass / gpuart_hip_render_direct (images are identical, bit for bit, in all modes):
 *   0 (default) fast: the launch-per-stage wavefront pipeline (per segment a persistent BVH-query launch that also carries
 *               the previous segment's Sun-shadow queries, and a shading launch; Sun-shadow queries stop at the first accepted
 *               hit and are skipped for surfaces facing away from the Sun), or — when the whole planned pass sequence is
 *               small (an interactive frame) — one persistent kernel per run (k_run) without the chain of dependent launches;
 *   1 "reference work": every query the reference performs is performed as a full closest-hit query, and exact counters
 *               are kept (gpuart_hip_counters); runs through k_run;
 *   2 megakernel: one thread runs a whole path (the first correct version; kept for A/B and cross-checks);
 *   3 the launch pipeline always;   5 k_run always;
 *   4 as 0, with counters of the work the fast mode really executes (rays, nodes, primitives); runs through k_run.
 * Direct lighting: 1 one thread per pixel with counters, 2 one thread per pixel, otherwise persistent lanes. */
int gpuart_hip_set_mode(gpuart_hip_ctx *ctx, int mode);
int gpuart_hip_counters(gpuart_hip_ctx *ctx, gpuart_counters *out, int reset);

/* HIP-event timing on the context's stream. Level 0: none; 1 (default): one event pair around every
 * render call (class 0); 2: additionally one pair around every BVH-query kernel (class 1).
 * gpuart_hip_kernel_time returns the sum and the count of class `cls` since its last reset. */
int gpuart_hip_set_timing(gpuart_hip_ctx *ctx, int level);
int gpuart_hip_kernel_time(gpuart_hip_ctx *ctx, int cls, double *total_ms, uint64_t *launches, int reset);

/* Scene statistics after upload: node count, primitive count, tree depth, device bytes. */
int gpuart_hip_scene_info(gpuart_hip_ctx *ctx, uint64_t *nodes, uint64_t *prims, uint32_t *max_depth,
                          uint64_t *device_bytes);

/* How the fast kernels walk the uploaded tree: 1 (the default for every tree of regular boxes since round 5) in the reference's order —
 * lower child first, shaders/bvh_intersection.glsl:432-441 — which is the only order PROVEN to return the reference's winner; 2 in the
 * reference's order with comparison-form box tests, because a box is irregular or does not bound its contents
 * (gpuart_hip_test_tree_class); 0 nearer child first with a certificate and a second walk where the certificate fails — ~10 % fewer node
 * visits, OPT-IN: gpuart_hip_set_nearest_first(ctx, min_prims) or GPUART_HIP_NEAREST_MIN_PRIMS=min_prims in the environment at
 * gpuart_hip_create (trees of at least min_prims primitives; 0xffffffff: never). Why opt-in: the reference accepts whatever parameter
 * its intersectors compute, and at grazing angles below ~1e-5 rad triangle.glsl:50-76 computes phantom hits far in front of the
 * triangle's own box; a walk that prunes that box after seeing a surface in between never tests the triangle and returns another winner
 * than the reference (tests/golden/order_adversary.npz: 16 constructed scenes, expected values from the reference's GLSL). No image of
 * a 3.7e11-ray soak of the benchmark scenes ever differed — the walk is soak-verified, not proven. */
int gpuart_hip_set_nearest_first(gpuart_hip_ctx *ctx, uint32_t min_prims);
int gpuart_hip_scene_order(gpuart_hip_ctx *ctx, int *order);

/* The test hooks (gpuart_hip_test_*: the run planner, the uploader's verdicts, the share-table check, birth orders, a stream stall, the
 * device functions one by one) are NOT part of this interface: include/gpuart_hip_test.h declares them, and only a library built with
 * -DGPUART_HIP_TEST_HOOKS (gpuart_amd/lib_test/, what the test suite loads) defines them. */

#ifdef __cplusplus
}
#endif
#endif /* GPUART_HIP_H */
