// capi.cpp — flat C API over the C++ library (see capi.h).
#include "capi.h"

#include <system_error>
#include <thread>
#include "exact_sort.h"

#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "renderer.h"
#include "scenes.h"
#include "utils.h"

using namespace gpuart;

namespace {

Primitive *make_primitive(const gpuart_prim_desc &d) {
    const float *f = d.f;
    switch (d.type) {
    case SPHERE: return new Sphere(Vec3f(f[0], f[1], f[2]), f[3]);
    case DISC: return new Disc(Vec3f(f[0], f[1], f[2]), Vec3f(f[3], f[4], f[5]), f[6]);
    case TRIANGLE: return new Triangle(Vec3f(f[0], f[1], f[2]), Vec3f(f[3], f[4], f[5]), Vec3f(f[6], f[7], f[8]));
    case CONE: return new Cone(Vec3f(f[0], f[1], f[2]), Vec3f(f[3], f[4], f[5]), f[6], f[7]);
    }
    return nullptr;
}

/// (Python hands over descriptions, the C++ API wants objects: for large scenes they are made and freed by several threads —
/// this is harness work, not part of SetPrimitives.)
template <class F>
void in_parts(size_t n, F body) {
    const size_t parts = n >= 65536 ? 8 : 1;
    std::vector<std::thread> th;
    struct Joiner {  // joins whatever was started, also when a part throws
        std::vector<std::thread> &th;
        ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
    } joiner{th};
    th.reserve(parts);
    for (size_t k = 1; k < parts; k++) {
        const size_t a = n * k / parts, b = n * (k + 1) / parts;
        try {
            th.emplace_back(body, a, b);
        } catch (const std::system_error &) {
            body(a, b);  // no thread to be had (the GPU box's CPU share caps them): this part runs here
        }
    }
    body(0, n / parts);
}

bool make_list(const gpuart_prim_desc *prims, int n, std::vector<Primitive *> &out) {
    const size_t base = out.size();
    out.resize(base + (size_t)n, nullptr);
    in_parts((size_t)n, [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) out[base + i] = make_primitive(prims[i]); });
    for (size_t i = base; i < out.size(); i++)
        if (!out[i]) return false;
    return true;
}

void free_list(std::vector<Primitive *> &v) {
    in_parts(v.size(), [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) delete v[i]; });
    v.clear();
}

double g_last_build_ms[2] = {0, 0};  ///< BoundingVolumesHierarchy's constructor, CompileTo — of the last compile_list (gpuart_last_build_ms)

int compile_list(std::vector<Primitive *> &list, unsigned maxLevels, unsigned minPrims, float **quads, size_t *nquads,
                 unsigned *depth) {
    const bool timing = std::getenv("GPUART_HOST_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    BoundingVolumesHierarchy tree(list, maxLevels, minPrims);
    const auto t1 = std::chrono::steady_clock::now();
    const size_t floats = tree.CompiledFloats();
    *quads = (float *)malloc(floats * sizeof(float) + 16);
    if (!*quads) return -1;
    tree.CompileTo(*quads);
    {
        const auto t2 = std::chrono::steady_clock::now();
        g_last_build_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        g_last_build_ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    }
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[gpuart] BVH of %zu primitives: build %.1f ms, compile %.1f ms\n", list.size(),
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    *nquads = floats / RGBA_ELEMS;
    if (depth) *depth = tree.GetDepth();
    return 0;
}

Camera make_camera(const float pos[3], const float dir[3], const float up[3], float fovY, float screenDist) {
    Camera c;
    c.Pos = Vec3f(pos[0], pos[1], pos[2]);
    c.Dir = Vec3f(dir[0], dir[1], dir[2]);
    c.Up = Vec3f(up[0], up[1], up[2]);
    c.FovY = fovY;
    c.ScreenDist = screenDist;
    return c;
}

}  // namespace

struct gpuart_renderer {
    Renderer impl;
    gpuart_renderer(unsigned w, unsigned h, const Camera &c, int device) : impl(w, h, c, device) {}
};

namespace {
template <typename T>
void vec3_ops(const T av[3], const T bv[3], T s, T out[36]) {
    const Vec3<T> a(av[0], av[1], av[2]), b(bv[0], bv[1], bv[2]);
    out[0] = a.length(); out[1] = a.sqrlength(); out[2] = a * b;
    const Vec3<T> r[11] = {a.normalized(), a ^ b, a + b, a - b, a * s, s * a, a / s, a.vrotx(s), a.vroty(s), a.vrotz(s), -a};
    for (int i = 0; i < 11; i++) { out[3 + 3 * i] = r[i].x; out[4 + 3 * i] = r[i].y; out[5 + 3 * i] = r[i].z; }
}
}  // namespace

extern "C" {

int gpuart_compile_bvh(const gpuart_prim_desc *prims, int n, unsigned maxLevels, unsigned minPrims, float **quads,
                       size_t *nquads, unsigned *depth) {
    if (!prims || n < 0 || !quads || !nquads) return -1;
    std::vector<Primitive *> list;
    int rc = make_list(prims, n, list) ? compile_list(list, maxLevels, minPrims, quads, nquads, depth) : -1;
    free_list(list);
    return rc;
}

int gpuart_compile_bvh_from_file(int kind, const char *path, float magnification, const float t[3],
                                 const gpuart_prim_desc *extra, int nextra, float **quads, size_t *nquads,
                                 unsigned *depth, size_t *nloaded) {
    if (!path || !t || !quads || !nquads) return -1;
    std::vector<Primitive *> list;
    const Vec3f tr(t[0], t[1], t[2]);
    bool ok = kind == 0 ? Utils::LoadMeshFromPLY(list, path, magnification, tr) : Utils::LoadPrimitives(list, path, magnification, tr);
    if (nloaded) *nloaded = list.size();
    int rc = -1;
    if (ok && make_list(extra, extra ? nextra : 0, list)) rc = compile_list(list, 1024, 2, quads, nquads, depth);
    free_list(list);
    return rc;
}

void gpuart_sort_permutation(const float *keys, size_t n, unsigned threads, uint32_t *perm) {
    std::vector<gpuart::SortKey> v(n);
    for (size_t i = 0; i < n; i++) v[i] = gpuart::SortKey{keys[i], (uint32_t)i};
    gpuart::ExactSort::Sort(v.data(), v.data() + n, threads);
    for (size_t i = 0; i < n; i++) perm[i] = v[i].index;
}

void gpuart_free(void *p) { free(p); }

void gpuart_last_build_ms(double out[2]) { out[0] = g_last_build_ms[0]; out[1] = g_last_build_ms[1]; }

void gpuart_camera_basis(const float pos[3], const float dir[3], const float up[3], float fovY, float screenDist,
                         unsigned width, unsigned height, float out[13]) {
    const Camera c = make_camera(pos, dir, up, fovY, screenDist);
    const Renderer::ScreenBasis s = Renderer::ComputeScreenBasis(c, width, height);
    s.Pos.storeIn(out); s.BottomLeft.storeIn(out + 3); s.DeltaHorz.storeIn(out + 6); s.DeltaVert.storeIn(out + 9);
    out[12] = Renderer::ComputePixelSize(c, height);
}

void gpuart_sun_direction(float azimuth, float altitude, float out[3]) {
    Renderer::ComputeSunDirection(azimuth, altitude).storeIn(out);
}

void gpuart_vec3f_ops(const float a[3], const float b[3], float s, float out[36]) { vec3_ops<float>(a, b, s, out); }
void gpuart_vec3d_ops(const double a[3], const double b[3], double s, double out[36]) { vec3_ops<double>(a, b, s, out); }

gpuart_renderer *gpuart_renderer_create(unsigned width, unsigned height, const float pos[3], const float dir[3],
                                        const float up[3], float fovY, float screenDist, int device) {
    return new gpuart_renderer(width, height, make_camera(pos, dir, up, fovY, screenDist), device);
}
void gpuart_renderer_destroy(gpuart_renderer *r) { delete r; }
int gpuart_renderer_is_ok(gpuart_renderer *r) { return r && r->impl.GetIsOK(); }

void gpuart_renderer_set_primitives(gpuart_renderer *r, const gpuart_prim_desc *prims, int n, int printInfo) {
    std::vector<Primitive *> list;
    if (make_list(prims, n, list)) r->impl.SetPrimitives(list, printInfo != 0);
    free_list(list);
}
void gpuart_renderer_init_box(gpuart_renderer *r) { InitBox(r->impl); }
int gpuart_renderer_init_dragon(gpuart_renderer *r, const char *plyPath) { return InitDragon(r->impl, plyPath) ? 1 : 0; }
int gpuart_renderer_init_cluster(gpuart_renderer *r, const char *datPath) {
    return (datPath ? InitCluster(r->impl, datPath) : InitCluster(r->impl)) ? 1 : 0;
}
int gpuart_renderer_init_tree(gpuart_renderer *r, const char *datPath) {
    return (datPath ? InitTree(r->impl, datPath) : InitTree(r->impl)) ? 1 : 0;
}
int gpuart_renderer_set_camera(gpuart_renderer *r, const float pos[3], const float dir[3], const float up[3], float fovY,
                               float screenDist) {
    return r->impl.SetCamera(make_camera(pos, dir, up, fovY, screenDist)) ? 1 : 0;
}
int gpuart_renderer_update_viewport(gpuart_renderer *r, unsigned w, unsigned h) { return r->impl.UpdateViewportSize(w, h) ? 1 : 0; }
int gpuart_renderer_set_tile(gpuart_renderer *r, unsigned x0, unsigned y0, unsigned w, unsigned h) {
    return r->impl.SetTile(x0, y0, w, h) ? 1 : 0;
}
int gpuart_renderer_set_interleaved_tile(gpuart_renderer *r, unsigned x0, unsigned y0, unsigned w, unsigned localRows,
                                         unsigned bandRows, unsigned bandStride) {
    return r->impl.SetInterleavedTile(x0, y0, w, localRows, bandRows, bandStride) ? 1 : 0;
}
void gpuart_renderer_set_sun(gpuart_renderer *r, float azimuth, float altitude, int directLighting) {
    r->impl.SetSunAzimuth(azimuth);
    r->impl.SetSunAltitude(altitude);
    r->impl.SetSunDirectLighting(directLighting != 0);
}
void gpuart_renderer_set_user_sphere(gpuart_renderer *r, const float pos[3], float radius, float emittance, int specular,
                                     int fuzzy) {
    r->impl.SetUserSphere(Vec3f(pos[0], pos[1], pos[2]), radius, emittance);
    r->impl.SetUserSphereSpecular(specular != 0);
    r->impl.SetUserSphereFuzzy(fuzzy != 0);
}
void gpuart_renderer_set_max_path_segments(gpuart_renderer *r, unsigned n) { r->impl.SetMaxPathSegments(n); }
void gpuart_renderer_set_seed(gpuart_renderer *r, uint32_t seed) { r->impl.SetSeed(seed); }
void gpuart_renderer_render_direct(gpuart_renderer *r) { r->impl.RenderDirectLighting(); }
void gpuart_renderer_restart_path_tracing(gpuart_renderer *r, unsigned perPass, unsigned perPixel) {
    r->impl.RestartPathTracing(perPass, perPixel);
}
unsigned gpuart_renderer_path_tracing_pass(gpuart_renderer *r) { return r->impl.RenderPathTracingPass(); }
int gpuart_renderer_read_direct(gpuart_renderer *r, float *rgba) { return r->impl.ReadDirectLighting(rgba) ? 1 : 0; }
int gpuart_renderer_read_radiance(gpuart_renderer *r, float *rgba, int normalized) {
    return r->impl.ReadRadiance(rgba, normalized != 0) ? 1 : 0;
}
int gpuart_renderer_finish(gpuart_renderer *r) { return r->impl.Finish() ? 1 : 0; }
int gpuart_renderer_save_checkpoint(gpuart_renderer *r, const char *path) { return r->impl.SaveCheckpoint(path) ? 1 : 0; }
int gpuart_renderer_load_checkpoint(gpuart_renderer *r, const char *path) { return r->impl.LoadCheckpoint(path) ? 1 : 0; }
gpuart_hip_ctx *gpuart_renderer_backend(gpuart_renderer *r) { return r->impl.GetBackend(); }
void gpuart_renderer_params(gpuart_renderer *r, gpuart_params *out) { *out = r->impl.MakeParams(); }
void gpuart_renderer_last_setprims_ms(gpuart_renderer *r, double out[4]) {
    for (int k = 0; k < 4; k++) out[k] = r->impl.GetLastSetPrimitivesMs()[k];
}

void gpuart_renderer_scene_info(gpuart_renderer *r, uint64_t *nodes, uint64_t *prims, unsigned *depth) {
    const BoundingVolumesHierarchy &t = r->impl.GetBVH();
    if (nodes) *nodes = t.GetNumNodes();
    if (prims) *prims = t.GetNumPrimitives();
    if (depth) *depth = t.GetDepth();
}

}  // extern "C"
