// renderer.cpp — host control of the hot path over the C-ABI device back end.
#include "renderer.h"

#include <algorithm>
#include <chrono>
#include <iomanip>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>

#include "utils.h"

#define GPUART_PI 3.1415926f  // the reference's PI (src/renderer.cpp:48); feeds tan() of the field of view

namespace gpuart {

// ---- pure host arithmetic (bit-compatible with the reference) -----------------------------------
// reference src/renderer.cpp:135-150
Renderer::ScreenBasis Renderer::ComputeScreenBasis(const Camera &cam, unsigned width, unsigned height) {
    const float aspect = (float)width / height;
    // cam.Up projected onto the plane orthogonal to cam.Dir
    const Vec3f up = ((cam.Dir ^ cam.Up) ^ cam.Dir).normalized();
    const Vec3f target = cam.Pos + cam.Dir.normalized() * cam.ScreenDist;
    // screen centre -> right edge, then centre -> top edge
    const Vec3f a = (cam.Dir.normalized() ^ up) * cam.ScreenDist * aspect * std::tan(cam.FovY / 2 * GPUART_PI / 180);
    const Vec3f b = up * a.length() / aspect;
    ScreenBasis s;
    s.Pos = cam.Pos;
    s.BottomLeft = target - a - b;
    s.DeltaHorz = 2 * a;
    s.DeltaVert = 2 * b;
    return s;
}

// reference src/renderer.cpp:573-574
float Renderer::ComputePixelSize(const Camera &cam, unsigned height) {
    return 2 * cam.ScreenDist * std::tan(cam.FovY / 2 * GPUART_PI / 180) / height;
}

// reference src/renderer.h:175-179
Vec3f Renderer::ComputeSunDirection(float azimuth, float altitude) {
    return Vec3f(1, 0, 0).vroty(-altitude).vrotz(azimuth);
}

// ---- lifecycle ----------------------------------------------------------------------------------
Renderer::Renderer(unsigned viewportWidth, unsigned viewportHeight, const Camera &camera, int device) {
    // defaults of the reference constructor (src/renderer.cpp:202-214)
    Lighting.azimuth = GPUART_PI;
    Lighting.altitude = GPUART_PI / 4;
    Lighting.directLightingEnabled = true;
    UserSphere.pos = Vec3f(0, 0, 0);
    UserSphere.emittance = 0;
    UserSphere.radius = 0;
    UserSphere.flags = 0;
    PathTracing.pathsPerPixel = 5;
    PathTracing.pathsPerPass = PathTracing.pathsPerPixel;
    PathTracing.numPathsRendered = 0;
    CurrentCamera = camera;

    if (!Check(gpuart_hip_create(device, &Backend), "creating the device back end")) return;
    if (viewportWidth == 0 || viewportHeight == 0) {
        std::cerr << "Renderer: viewport must not be empty." << std::endl;
        return;
    }
    IsOK = true;  // UpdateViewportSize reports through IsOK
    UpdateViewportSize(viewportWidth, viewportHeight);
}

Renderer::~Renderer() {
    if (Backend) gpuart_hip_destroy(Backend);
}

bool Renderer::Check(int status, const char *what) {
    if (status == 0) return true;
    std::cerr << "Renderer: error " << status << " while " << what << ": " << gpuart_hip_last_error() << std::endl;
    return false;
}

bool Renderer::UpdateViewportSize(unsigned width, unsigned height) {
    if (!Backend || width == 0 || height == 0) return IsOK = false;
    Viewport.width = width;
    Viewport.height = height;
    if (!Check(gpuart_hip_resize(Backend, width, height), "allocating per-pixel buffers")) return IsOK = false;
    Tile.x = Tile.y = 0; Tile.w = width; Tile.h = height;
    if (!SetCamera(CurrentCamera)) IsOK = false;
    return IsOK;
}

bool Renderer::SetTile(unsigned x0, unsigned y0, unsigned w, unsigned h) {
    if (!Backend) return false;
    if (!Check(gpuart_hip_set_tile(Backend, x0, y0, w, h), "setting the tile")) return false;
    Tile.x = x0; Tile.y = y0; Tile.w = w; Tile.h = h;
    ResetPathTracing();
    return true;
}

bool Renderer::SetInterleavedTile(unsigned x0, unsigned y0, unsigned w, unsigned localRows, unsigned bandRows,
                                  unsigned bandStride) {
    if (!Backend) return false;
    if (!Check(gpuart_hip_set_tile_interleaved(Backend, x0, y0, w, localRows, bandRows, bandStride), "setting the tile"))
        return false;
    Tile.x = x0; Tile.y = y0; Tile.w = w; Tile.h = localRows;
    ResetPathTracing();
    return true;
}

bool Renderer::SetNearestFirst(uint32_t minPrims) {
    if (!Backend) return false;
    if (!Check(gpuart_hip_set_nearest_first(Backend, minPrims), "choosing the visiting order")) return false;
    ResetPathTracing();
    return true;
}

bool Renderer::SetShare(int rank, int nranks) {
    if (!Backend) return false;
    gpuart_tile_geom g;
    if (gpuart_hip_share_of_rank(Viewport.width, Viewport.height, rank, nranks, 8, &g) != 0 || g.th == 0) {
        std::cerr << "Renderer: no share " << rank << " of " << nranks << " in a frame of " << Viewport.height << " rows." << std::endl;
        return false;
    }
    return SetInterleavedTile(g.x0, g.y0, g.tw, g.th, g.band_rows, g.band_stride);
}

/// Bound of one phase of the multi-GPU read-out for the library's watchdog (gpuart_hip_phase_begin): GPUART_PHASE_TIMEOUT_MS,
/// default 300 s — above the library's own bounds (GPUART_HIP_COMM_TIMEOUT_MS 120 s, GPUART_HIP_GATHER_TIMEOUT_MS 60 s), which
/// come back with an error first; the watchdog is for whatever those do not wrap.
static uint32_t PhaseTimeoutMs() {
    const char *v = getenv("GPUART_PHASE_TIMEOUT_MS");
    if (!v) return 300000u;
    const long x = strtol(v, nullptr, 10);
    return x <= 0 ? 0u : (uint32_t)x;
}

bool Renderer::GatherRadiance(Renderer *const *ranks, int n, int root, bool normalized, float *fullFrame) {
    if (!ranks || n < 1 || root < 0 || root >= n || !fullFrame) return false;
    std::vector<gpuart_hip_ctx *> ctxs((size_t)n);
    for (int k = 0; k < n; k++) {
        if (!ranks[k] || !ranks[k]->IsOK) return false;
        ctxs[(size_t)k] = ranks[k]->Backend;
    }
    Renderer &r0 = *ranks[root];
    const float div = normalized && r0.PathTracing.numPathsRendered ? (float)r0.PathTracing.numPathsRendered : 1.0f;
    // One communicator per set of renderers, kept by the contexts themselves. Whether these contexts are (still) the ranks
    // 0..n-1 of one is the library's to say: it answers GPUART_HIP_ERR_NO_COMM before anything is transferred, then one is made.
    // Every step that waits for RCCL or for the other GPUs is a named phase (a line on stderr before and after, and the
    // library's watchdog behind it): a read-out that stalls says where.
    const uint32_t bound = PhaseTimeoutMs();
    gpuart_hip_phase_begin("frame gather (gpuart_hip_gather_all_read)", bound);
    int rc = gpuart_hip_gather_all_read(ctxs.data(), n, 1, div, root, fullFrame);
    gpuart_hip_phase_end();
    if (rc == GPUART_HIP_ERR_NO_COMM) {
        gpuart_hip_phase_begin("communicator init (gpuart_hip_comm_init_all = ncclCommInitAll)", bound);
        const bool made = r0.Check(gpuart_hip_comm_init_all(ctxs.data(), n), "creating the RCCL communicator");
        gpuart_hip_phase_end();
        if (!made) return false;
        gpuart_hip_phase_begin("frame gather (gpuart_hip_gather_all_read)", bound);
        rc = gpuart_hip_gather_all_read(ctxs.data(), n, 1, div, root, fullFrame);
        gpuart_hip_phase_end();
    }
    return r0.Check(rc, "gathering the frame");
}

bool Renderer::ReleaseCommunicator(Renderer *const *ranks, int n) {
    if (!ranks || n < 1) return false;
    bool ok = true;
    gpuart_hip_phase_begin("communicator destroy (gpuart_hip_comm_destroy = ncclCommDestroy)", PhaseTimeoutMs());
    for (int k = 0; k < n; k++)
        if (ranks[k] && ranks[k]->Backend) ok = ranks[k]->Check(gpuart_hip_comm_destroy(ranks[k]->Backend), "destroying the RCCL communicator") && ok;
    gpuart_hip_phase_end();
    return ok;
}

bool Renderer::SetCamera(const Camera &cam) {
    CurrentCamera = cam;
    if (!Backend) return false;
    const ScreenBasis s = ComputeScreenBasis(cam, Viewport.width, Viewport.height);
    float pos[3], bl[3], dh[3], dv[3];
    s.Pos.storeIn(pos); s.BottomLeft.storeIn(bl); s.DeltaHorz.storeIn(dh); s.DeltaVert.storeIn(dv);
    if (!Check(gpuart_hip_set_camera(Backend, pos, bl, dh, dv), "setting the camera")) return false;
    ResetPathTracing();
    return true;
}

// ---- scene --------------------------------------------------------------------------------------
namespace {
struct ByteCount {
    size_t count;
};
std::ostream &operator<<(std::ostream &os, const ByteCount &bc) {
    static const char *unit[] = {" B", " KiB", " MiB", " GiB"};
    double v = (double)bc.count;
    int u = 0;
    while (v >= 1024 && u < 3) { v /= 1024; u++; }
    return os << std::fixed << std::setprecision(u ? 1 : 0) << v << unit[u];
}
}  // namespace

void Renderer::SetPrimitives(std::vector<Primitive *> &primitives, bool printInfo) {
    auto t0 = std::chrono::high_resolution_clock::now();
    const auto tAll = t0;
    if (printInfo) std::cout << "Constructing BVH tree of " << primitives.size() << " primitives... " << std::flush;
    Tree = BoundingVolumesHierarchy(primitives, 1024, 2);  // reference src/renderer.cpp:454
    const auto tBuilt = std::chrono::high_resolution_clock::now();
    if (printInfo) {
        std::cout << "done (" << Utils::TimeElapsed(t0) << ")." << std::endl;
        std::cout << "Compiling BVH tree... " << std::flush;
        t0 = std::chrono::high_resolution_clock::now();
    }
    // (the reference compiles into a Primitive::Data — src/renderer.cpp:460-466 —; the same quads go into a buffer that is not zeroed
    // first: for an 871 200-triangle mesh that is 105 MB one thread would touch before the threads that fill it)
    const size_t compiledFloats = Tree.CompiledFloats();
    std::unique_ptr<float[]> compiled(new float[compiledFloats]);
    Tree.CompileTo(compiled.get());
    const auto tCompiled = std::chrono::high_resolution_clock::now();
    if (printInfo) std::cout << "done (" << Utils::TimeElapsed(t0) << ").\n";
    if (!Backend || !Check(gpuart_hip_upload_bvh(Backend, compiled.get(), compiledFloats / RGBA_ELEMS), "uploading the BVH"))
        IsOK = false;
    if (printInfo) std::cout << "Compiled tree occupies " << ByteCount{compiledFloats * sizeof(float)} << "." << std::endl;
    const auto tEnd = std::chrono::high_resolution_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    LastSetPrimitivesMs[0] = ms(tAll, tEnd); LastSetPrimitivesMs[1] = ms(tAll, tBuilt); LastSetPrimitivesMs[2] = ms(tBuilt, tCompiled);
    LastSetPrimitivesMs[3] = ms(tCompiled, tEnd);
    if (std::getenv("GPUART_HOST_TIMING")) {
        fprintf(stderr, "[gpuart] Renderer::SetPrimitives(%zu primitives): %.1f ms (build %.1f + compile %.1f + re-layout and upload %.1f)\n",
                primitives.size(), ms(tAll, tEnd), ms(tAll, tBuilt), ms(tBuilt, tCompiled), ms(tCompiled, tEnd));
    }
    ResetPathTracing();
}

// ---- lighting / user sphere ---------------------------------------------------------------------
void Renderer::SetUserSphere(const Vec3f &pos, float radius, float emittance) {
    UserSphere.pos = pos;
    UserSphere.radius = radius;
    SetUserSphereEmittance(emittance);
}

void Renderer::SetUserSphereEmittance(float em) {
    UserSphere.emittance = em;
    SetFlag(EM_NONZERO, em > 0);
}

void Renderer::SetFlag(uint32_t flag, bool on) {
    if (on) UserSphere.flags |= flag;
    else UserSphere.flags &= ~flag;
    ResetPathTracing();
}

gpuart_params Renderer::MakeParams() const {
    gpuart_params p{};
    const Vec3f sun = ComputeSunDirection(Lighting.azimuth, Lighting.altitude);
    p.sunDirAlt[0] = sun.x; p.sunDirAlt[1] = sun.y; p.sunDirAlt[2] = sun.z; p.sunDirAlt[3] = Lighting.altitude;
    p.sunEnabled = Lighting.directLightingEnabled ? 1 : 0;
    p.userSphere[0] = UserSphere.pos.x; p.userSphere[1] = UserSphere.pos.y; p.userSphere[2] = UserSphere.pos.z;
    p.userSphere[3] = UserSphere.radius;
    const Vec3f em = Vec3f(1, 1, 1) * UserSphere.emittance;
    p.userSphereEm[0] = em.x; p.userSphereEm[1] = em.y; p.userSphereEm[2] = em.z;
    p.userSphereFlags = UserSphere.flags;
    p.pixelSize = ComputePixelSize(CurrentCamera, Viewport.height);
    p.cameraPos[0] = CurrentCamera.Pos.x; p.cameraPos[1] = CurrentCamera.Pos.y; p.cameraPos[2] = CurrentCamera.Pos.z;
    p.maxSegments = (int32_t)MaxPathSegments;
    p.minWeight = MinWeight;
    return p;
}

// ---- rendering ----------------------------------------------------------------------------------
void Renderer::RenderDirectLighting() {
    if (!IsOK) return;
    const gpuart_params p = MakeParams();
    Check(gpuart_hip_render_direct(Backend, &p), "rendering direct lighting");
}

void Renderer::ResetPathTracing() {
    PathTracing.numPathsRendered = 0;
    if (Backend && Viewport.width) {
        gpuart_hip_pt_reset(Backend);
        // the passes RenderPathTracingPass() will submit until pathsPerPixel is reached (a scheduling hint)
        const unsigned per = PathTracing.pathsPerPass ? PathTracing.pathsPerPass : 1;
        gpuart_hip_pt_plan(Backend, (PathTracing.pathsPerPixel + per - 1) / per);
    }
}

void Renderer::RestartPathTracing(unsigned pathsPerPass, unsigned pathsPerPixel) {
    PathTracing.pathsPerPixel = pathsPerPixel;
    PathTracing.pathsPerPass = std::min(pathsPerPass, pathsPerPixel);
    ResetPathTracing();
}

void Renderer::ExtendPathTracing(unsigned pathsPerPass, unsigned pathsPerPixel) {
    PathTracing.pathsPerPixel = std::max(pathsPerPixel, PathTracing.numPathsRendered);
    PathTracing.pathsPerPass = std::max(1u, std::min(pathsPerPass, PathTracing.pathsPerPixel));
    if (Backend && Viewport.width) {
        const unsigned left = PathTracing.pathsPerPixel - PathTracing.numPathsRendered;
        gpuart_hip_pt_plan(Backend, (left + PathTracing.pathsPerPass - 1) / PathTracing.pathsPerPass);
    }
}

unsigned Renderer::RenderPathTracingPass() {
    if (!IsOK) return PathTracing.numPathsRendered;
    if (PathTracing.numPathsRendered < PathTracing.pathsPerPixel) {
        const unsigned pathsToRender =
            std::min(PathTracing.pathsPerPass, PathTracing.pathsPerPixel - PathTracing.numPathsRendered);
        const gpuart_params p = MakeParams();
        // RandSeed: four draws per pass from the never re-seeded generator (reference src/renderer.cpp:585-589)
        std::uniform_real_distribution<float> distr(0, 1);
        float seed[4];
        for (float &s : seed) s = distr(RndGen);
        if (Check(gpuart_hip_pt_pass(Backend, &p, seed, (int)pathsToRender), "rendering a path-tracing pass"))
            PathTracing.numPathsRendered += pathsToRender;
    }
    return PathTracing.numPathsRendered;
}

bool Renderer::ReadDirectLighting(float *rgba) {
    return IsOK && Check(gpuart_hip_read(Backend, 0, rgba, 1.0f), "reading the frame");
}

bool Renderer::ReadRadiance(float *rgba, bool normalized) {
    // the division is the reference's ptracingNormalize program (shaders/pt_normalize.glsl:44-47)
    const float div = normalized && PathTracing.numPathsRendered ? (float)PathTracing.numPathsRendered : 1.0f;
    return IsOK && Check(gpuart_hip_read(Backend, 1, rgba, div), "reading the radiance accumulator");
}

// ---- checkpoint / resume ---------------------------------------------------------------------------------------------
namespace {
const char CK_MAGIC[8] = {'G', 'P', 'U', 'A', 'R', 'T', 'C', 'K'};
struct CkHeader {
    char magic[8];
    uint32_t version, width, height, tileX, tileY, tileW, tileH;
    uint32_t numPathsRendered, pathsPerPixel, pathsPerPass, maxPathSegments;
    float minWeight;
    uint32_t rngTextBytes;
};
}  // namespace

bool Renderer::SaveCheckpoint(const char *fileName) {
    if (!IsOK) return false;
    std::vector<float> acc((size_t)Tile.w * Tile.h * 4);
    if (!ReadRadiance(acc.data(), false)) return false;
    std::ostringstream rng;
    rng << RndGen;  // the full mt19937 state, as text
    const std::string rngText = rng.str();
    CkHeader h{};
    memcpy(h.magic, CK_MAGIC, 8);
    h.version = 1; h.width = Viewport.width; h.height = Viewport.height;
    h.tileX = Tile.x; h.tileY = Tile.y; h.tileW = Tile.w; h.tileH = Tile.h;
    h.numPathsRendered = PathTracing.numPathsRendered; h.pathsPerPixel = PathTracing.pathsPerPixel;
    h.pathsPerPass = PathTracing.pathsPerPass; h.maxPathSegments = MaxPathSegments; h.minWeight = MinWeight;
    h.rngTextBytes = (uint32_t)rngText.size();
    std::ofstream f(fileName, std::ios::binary);
    f.write((const char *)&h, sizeof h);
    f.write(rngText.data(), (std::streamsize)rngText.size());
    f.write((const char *)acc.data(), (std::streamsize)(acc.size() * sizeof(float)));
    return f.good();
}

bool Renderer::LoadCheckpoint(const char *fileName) {
    if (!IsOK) return false;
    std::ifstream f(fileName, std::ios::binary);
    CkHeader h{};
    f.read((char *)&h, sizeof h);
    if (!f.good() || memcmp(h.magic, CK_MAGIC, 8) != 0 || h.version != 1) {
        std::cerr << "Renderer: \"" << fileName << "\" is not a checkpoint." << std::endl;
        return false;
    }
    if (h.width != Viewport.width || h.height != Viewport.height || h.tileX != Tile.x || h.tileY != Tile.y ||
        h.tileW != Tile.w || h.tileH != Tile.h || h.rngTextBytes > (1u << 20)) {
        std::cerr << "Renderer: checkpoint does not match the current viewport / tile." << std::endl;
        return false;
    }
    std::string rngText(h.rngTextBytes, '\0');
    f.read(&rngText[0], (std::streamsize)rngText.size());
    std::vector<float> acc((size_t)Tile.w * Tile.h * 4);
    f.read((char *)acc.data(), (std::streamsize)(acc.size() * sizeof(float)));
    if (!f.good()) return false;
    std::istringstream rng(rngText);
    std::mt19937 gen;
    rng >> gen;
    if (rng.fail()) return false;
    if (!Check(gpuart_hip_write(Backend, 1, acc.data()), "restoring the radiance accumulator")) return false;
    RndGen = gen;
    PathTracing.numPathsRendered = h.numPathsRendered;
    PathTracing.pathsPerPixel = h.pathsPerPixel;
    PathTracing.pathsPerPass = h.pathsPerPass;
    MaxPathSegments = h.maxPathSegments;
    MinWeight = h.minWeight;
    return true;
}

bool Renderer::Finish() { return Backend && Check(gpuart_hip_finish(Backend), "waiting for the device"); }

}  // namespace gpuart
