/* capi.h — flat C API over the C++ Renderer/Scene library (libgpuart.so), used by the Python
 * plumbing (ctypes) in tests and bench.py. Thin by design: one call per Renderer method. */
#ifndef GPUART_CAPI_H
#define GPUART_CAPI_H

#include <stddef.h>
#include <stdint.h>

#include "gpuart_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A primitive description: sphere {c r}, disc {c n r}, triangle {v0 v1 v2}, cone {c1 c2 r1 r2}. */
typedef struct gpuart_prim_desc {
    int32_t type;
    float f[9];
} gpuart_prim_desc;

typedef struct gpuart_renderer gpuart_renderer;

/* ---- pure host functions (no device needed) ---- */
/* BoundingVolumesHierarchy(prims, maxLevels, minPrims).Compile(); free *quads with gpuart_free. */
int gpuart_compile_bvh(const gpuart_prim_desc *prims, int n, unsigned maxLevels, unsigned minPrims, float **quads,
                       size_t *nquads, unsigned *depth);
/* The same for a scene file: kind 0 = ASCII PLY mesh (Utils::LoadMeshFromPLY), 1 = primitive list
 * (Utils::LoadPrimitives); `extra` primitives are appended after the loaded ones. */
int gpuart_compile_bvh_from_file(int kind, const char *path, float magnification, const float translation[3],
                                 const gpuart_prim_desc *extra, int nextra, float **quads, size_t *nquads,
                                 unsigned *depth, size_t *nloaded);
void gpuart_free(void *p);
/* What the library itself spent in the last gpuart_compile_bvh / _from_file of this process: out[0] = BoundingVolumesHierarchy's
 * constructor (the build), out[1] = its compilation into quads, in ms — without the harness around them (making and deleting one
 * Primitive object per description, copying the result into the caller's language). */
void gpuart_last_build_ms(double out[2]);
/* The permutation the BVH build applies to a node's primitives: perm[i] = position (before sorting) of the element that
 * std::sort leaves at i when sorting by `keys` with operator< — computed by exact_sort.h on up to `threads` threads. */
void gpuart_sort_permutation(const float *keys, size_t n, unsigned threads, uint32_t *perm);
/* out[13] = Pos(3) BottomLeft(3) DeltaHorz(3) DeltaVert(3) PixelSize */
void gpuart_camera_basis(const float pos[3], const float dir[3], const float up[3], float fovY, float screenDist,
                         unsigned width, unsigned height, float out[13]);
void gpuart_sun_direction(float azimuth, float altitude, float out[3]);
/// Every Vec3<float> / Vec3<double> operation of math_types.h once (parity hook: tests/golden/host_math.npz holds what the
/// reference's own src/math_types.h computes): out = {length, sqrlength, a*b} + normalized + a^b + a+b + a-b + a*s + s*a + a/s +
/// vrotx(s) + vroty(s) + vrotz(s) + (-a).
void gpuart_vec3f_ops(const float a[3], const float b[3], float s, float out[36]);
void gpuart_vec3d_ops(const double a[3], const double b[3], double s, double out[36]);

/* ---- gpuart::Renderer ---- */
gpuart_renderer *gpuart_renderer_create(unsigned width, unsigned height, const float pos[3], const float dir[3],
                                        const float up[3], float fovY, float screenDist, int device);
void gpuart_renderer_destroy(gpuart_renderer *r);
int gpuart_renderer_is_ok(gpuart_renderer *r);
void gpuart_renderer_set_primitives(gpuart_renderer *r, const gpuart_prim_desc *prims, int n, int printInfo);
void gpuart_renderer_init_box(gpuart_renderer *r);
int gpuart_renderer_init_dragon(gpuart_renderer *r, const char *plyPath);
/* InitCluster / InitTree (reference src/scenes.cpp:69-103) on a primitive-list file; NULL = the reference's own path. */
int gpuart_renderer_init_cluster(gpuart_renderer *r, const char *datPath);
int gpuart_renderer_init_tree(gpuart_renderer *r, const char *datPath);
int gpuart_renderer_set_camera(gpuart_renderer *r, const float pos[3], const float dir[3], const float up[3], float fovY,
                               float screenDist);
int gpuart_renderer_update_viewport(gpuart_renderer *r, unsigned width, unsigned height);
int gpuart_renderer_set_tile(gpuart_renderer *r, unsigned x0, unsigned y0, unsigned w, unsigned h);
int gpuart_renderer_set_interleaved_tile(gpuart_renderer *r, unsigned x0, unsigned y0, unsigned w, unsigned localRows,
                                         unsigned bandRows, unsigned bandStride);
void gpuart_renderer_set_sun(gpuart_renderer *r, float azimuth, float altitude, int directLighting);
void gpuart_renderer_set_user_sphere(gpuart_renderer *r, const float pos[3], float radius, float emittance, int specular,
                                     int fuzzy);
void gpuart_renderer_set_max_path_segments(gpuart_renderer *r, unsigned n);
void gpuart_renderer_set_seed(gpuart_renderer *r, uint32_t seed);
void gpuart_renderer_render_direct(gpuart_renderer *r);
void gpuart_renderer_restart_path_tracing(gpuart_renderer *r, unsigned pathsPerPass, unsigned pathsPerPixel);
unsigned gpuart_renderer_path_tracing_pass(gpuart_renderer *r);
int gpuart_renderer_read_direct(gpuart_renderer *r, float *rgba);
int gpuart_renderer_read_radiance(gpuart_renderer *r, float *rgba, int normalized);
int gpuart_renderer_finish(gpuart_renderer *r);
int gpuart_renderer_save_checkpoint(gpuart_renderer *r, const char *path);
int gpuart_renderer_load_checkpoint(gpuart_renderer *r, const char *path);
gpuart_hip_ctx *gpuart_renderer_backend(gpuart_renderer *r);
void gpuart_renderer_params(gpuart_renderer *r, gpuart_params *out);
/* What the renderer's last SetPrimitives spent inside the library, ms: whole call, BVH build, compilation, re-layout + upload. */
void gpuart_renderer_last_setprims_ms(gpuart_renderer *r, double out[4]);
void gpuart_renderer_scene_info(gpuart_renderer *r, uint64_t *nodes, uint64_t *prims, unsigned *depth);

#ifdef __cplusplus
}
#endif
#endif
