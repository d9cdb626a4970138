/* gpuart_hip_test.h — test hooks of libgpuart_hip.so. NOT part of the product's C ABI (include/gpuart_hip.h): these entry points exist
 * only in a library compiled with -DGPUART_HIP_TEST_HOOKS — gpuart_amd/lib_test/libgpuart_hip.so, built beside the product library
 * from the same sources (csrc/Makefile) and loaded by the test suite (tests/conftest.py sets GPUART_LIBDIR). The product library
 * gpuart_amd/lib/libgpuart_hip.so — what libgpuart.so, gpuart_cli, bench.py and smoke() run on — neither defines nor exports them
 * (tests/test_host_parity.py::test_c_abi_exports_every_declared_symbol holds both libraries to their headers).
 * The parity tests call the device code function by function through these; each mirrors one reference GLSL function or one
 * host-side decision of the library. */
#ifndef GPUART_HIP_TEST_H
#define GPUART_HIP_TEST_H

#include "gpuart_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook of the mechanism behind the bounded RCCL calls (no device needed): a call that holds for hold_ms under a bound of
 * timeout_ms (0: inline, unbounded). 0 if it returned in time, GPUART_HIP_ERR_TIMEOUT otherwise; mark_stuck != 0 leaves the
 * communicator layer marked out of service as a real timeout does. */
int gpuart_hip_test_bounded_call(uint32_t hold_ms, uint32_t timeout_ms, int mark_stuck);


/* ---- run planner test hook (pure host code: needs no device and no context) ------------------
 * Drives the library's own run planner (csrc/hip/run_planner.h — the code gpuart_hip_resize / _set_share / _pt_plan /
 * _set_mode / _pt_pass / _flush take their scheduling decisions from) through a sequence of operations and reports every
 * pipeline run it would launch, so that a test can assert the invariants without a GPU: every run has 1 <= passes <=
 * max_batch (a lane's path buffers hold n_slots x max_batch paths), stays within the path budget, and passes are launched in
 * order, each exactly once.
 *   cfg[8]  = { max_batch cap (1..64), pass lanes, batch Mpaths, min-run Kpaths, small Kpaths, lane budget MB,
 *               plan-run percent, 0 }                       (0 in a field = the library's default)
 *   ops     = n_ops triples (op, a, b):  0 RESIZE(width, height)   1 SHARE(rank, nranks) of the current frame, 8-row bands
 *             2 PLAN(passes, -)   3 MODE(mode, -)   4 PASS(count, -) = count calls of pt_pass   5 FLUSH(-, -)
 *             6 ALLOC_FAILS(n, -): the next RESIZE / SHARE finds the device refusing its first n allocation attempts
 *   runs    = up to max_runs records of 6 words: { index of the op that launched it, n_slots, max_batch, passes of the run,
 *             1 if it goes through k_run, passes still pending afterwards }
 * Returns the number of runs (which may exceed max_runs: only the first max_runs are stored), or a negative error. */
int gpuart_hip_test_planner(const uint32_t cfg[8], const uint32_t *ops, int n_ops, uint32_t *runs, int max_runs);

/* What gpuart_hip_upload_bvh decides about a canonical compiled tree (pure host code: needs no device and no context): *flags =
 * bit 0: some box is irregular (min > max, NaN or infinite on an axis) — box tests take the reference's comparison form;
 * bit 1: some box does not bound what it holds by the reference's own formulas (a child outside its parent, a primitive outside its
 *        leaf's box, a negative radius, cone constants that contradict its centres, coordinates beyond 2^20) — with either bit the
 *        tree is walked in the reference's order throughout; with neither the fast kernels visit the nearer child first;
 * bit 2: some box plane lies within 2^-60 of zero without being zero, subnormal numbers included (no quick box answers: below);
 * bits 8-11: the primitive types present. GPUART_HIP_ERR_ARG for a malformed tree (gpuart_hip_last_error says why). */
int gpuart_hip_test_tree_class(const float *quads, size_t nquads, uint32_t *flags);

/* The slack constant the upload assigns to a tree for the quick box answers of the BVH queries (csrc/hip/box_quick.h: the slab entry
 * of a box, taken where margins prove it equal to the reference's six face tests — IntersectsAABB, shaders/bvh_intersection.glsl:229-354 —
 * bit for bit): 4 * 2^-24 * (largest |plane coordinate| of the root's box) + 2^-90, or +inf — every box test runs its six face tests —
 * for a tree with bit 0, 1 or 2 of gpuart_hip_test_tree_class set (and, on a context, with GPUART_HIP_QUICK_BOXES=0). Pure host code. */
int gpuart_hip_test_tree_slack(const float *quads, size_t nquads, float *slack);

/* The validation every rank of gpuart_hip_gather applies to the exchanged share table (pure host code): `shares[k]` and
 * `status[k]` (0: ready) as rank k announced them. 0 if the gather would go ahead — full-width rows, every frame row covered
 * exactly once, every rank ready — else the error code it would return on every rank (gpuart_hip_last_error says why). */
int gpuart_hip_test_share_table(const gpuart_tile_geom *shares, const uint32_t *status, int n, int which, int root);

/* The order in which the paths of a pass are BORN: path slots [64 k, 64 k + 64) hold the 8x8 pixel block order[k] of the context's
 * tile (blocks numbered row-major, ceil(tw / 8) per row; `n` = their number). No pixel's value depends on it — a path's random
 * numbers come from hit positions and the pass's seed (path_tracing.glsl:164-165, 220) — which the parity tests check with
 * arbitrary permutations. order == NULL: back to row-major. Dropped when the tile or frame size changes. */
int gpuart_hip_test_tile_order(gpuart_hip_ctx *ctx, const uint32_t *order, size_t n);

/* The library's own choice of that order (runs of one pass through the persistent kernel, GPUART_HIP_TILE_ORDER=0 turns it off): the
 * first such run after a change of tile, camera or scene counts the shaded path segments per block, a device sort behind it puts the
 * blocks into classes of that count, most expensive class first and row-major within a class, and the runs that follow are born in
 * that order. This hook runs the sort alone: `cost[n]` -> `order[n]` (a permutation of 0 .. n-1). */
int gpuart_hip_test_sort_tiles(gpuart_hip_ctx *ctx, const uint32_t *cost, size_t n, uint32_t *order);
/* The order the next run of one pass would be born in (after everything launched so far has finished): 1 and `order[n]` filled, or
 * 0 if there is none yet (row-major); < 0 on error. */
int gpuart_hip_test_current_tile_order(gpuart_hip_ctx *ctx, uint32_t *order, size_t n);

/* Keeps the context's primary stream busy for `ms` milliseconds (one idle-spinning wave that ends by itself; at most 5000):
 * what a peer that has not arrived looks like to the bounded waits of gpuart_hip_gather / gpuart_hip_wait, on one GPU. */
int gpuart_hip_test_stall(gpuart_hip_ctx *ctx, uint32_t ms);

/* ---- device-function test hooks (parity tests call the device code through these) ----------
 * Arrays are n x 4 float32 in host memory. Each mirrors one reference GLSL function. */
int gpuart_hip_test_random(gpuart_hip_ctx *ctx, const float *in, int n, float *out);
int gpuart_hip_test_math(gpuart_hip_ctx *ctx, const float *in, int n, float *out); /* sin, cos, pow(y,16), sqrt(y) */
int gpuart_hip_test_hemisphere(gpuart_hip_ctx *ctx, const float *v, const float *ri, int n, float *out);
int gpuart_hip_test_inside_cone(gpuart_hip_ctx *ctx, const float *v, const float *normal, const float *ri,
                                float halfAngle, int n, float *out);
int gpuart_hip_test_sky(gpuart_hip_ctx *ctx, const float *dir, const float sunDirAlt[4], int n, float *out);
/* Intersectors: `quads` = n canonical primitive payloads (StoreDataIntoBVH layout, 4 quads per
 * primitive, unused quads ignored); out0 = (pos, P), out1 = (N, 0), zeros after pos on a miss. */
int gpuart_hip_test_intersect(gpuart_hip_ctx *ctx, int ptype, const float *rs, const float *rd, const float *quads,
                              int n, float *out0, float *out1);
int gpuart_hip_test_aabb(gpuart_hip_ctx *ctx, const float *rs, const float *rd, const float *bmin, const float *bmax,
                         int n, float *out);
/* Closest-hit query over the uploaded tree incl. the user sphere; out0 = (pos, P), out1 = (N, type
 * (+0.5 if the user sphere was hit), or -1 on a miss). any_hit == 1: out0[0] = 1/0 only. any_hit == 2: the closest-hit
 * query in the order of the fast kernels (nearer child first, lower primitive index wins equal parameters; trees of regular
 * boxes only) — it must return what the reference's order returns. */
int gpuart_hip_test_traverse(gpuart_hip_ctx *ctx, const float *rs, const float *rd, const float userSphere[4], int n,
                             int any_hit, float *out0, float *out1);
/* Camera rays of the context's tile: rstart / rdir, tw*th*4 floats each. */
int gpuart_hip_test_cam_rays(gpuart_hip_ctx *ctx, float *rstart, float *rdir);

#ifdef __cplusplus
}
#endif
#endif
