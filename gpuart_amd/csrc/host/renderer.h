// renderer.h — gpuart::Camera and gpuart::Renderer: the reference's public API
// (src/renderer.h:39-49,183-296), re-hosted on the MI355X back end (include/gpuart_hip.h)
// instead of OpenGL. Every reference method keeps its name, arguments and behaviour
// (including: every setter silently restarts the path-tracing accumulation; failures are
// reported through bool returns / GetIsOK() and std::cerr, never exceptions).
//
// Additions (the reference draws into the GL framebuffer, which no longer exists, and has
// MAX_PATH_SEGMENTS / the RNG seed / the viewport fixed at compile time):
//   ReadDirectLighting, ReadRadiance, Finish, SetMaxPathSegments, SetMinWeight, SetSeed,
//   SetTile, GetBackend, ComputeScreenBasis, GetNumPathsRendered.
#ifndef GPUART_RENDERER_H
#define GPUART_RENDERER_H

#include <cstdint>
#include <random>
#include <vector>

#include "bvh.h"
#include "core.h"
#include "gpuart_hip.h"
#include "math_types.h"

namespace gpuart {

class Camera {
public:
    Vec3f Pos;         ///< camera position
    Vec3f Dir;         ///< viewing direction
    Vec3f Up;          ///< "up" direction
    float FovY;        ///< vertical field of view, degrees
    float ScreenDist;  ///< distance from Pos to the virtual screen the rays start on
};

class Renderer {
public:
    /// The four uniforms of the reference's cameraInit program (src/renderer.cpp:135-166).
    struct ScreenBasis {
        Vec3f Pos, BottomLeft, DeltaHorz, DeltaVert;
    };
    static ScreenBasis ComputeScreenBasis(const Camera &cam, unsigned width, unsigned height);
    /// PixelSize uniform (src/renderer.cpp:573-574).
    static float ComputePixelSize(const Camera &cam, unsigned height);
    /// SunDirAlt.xyz (src/renderer.h:175-179).
    static Vec3f ComputeSunDirection(float azimuth, float altitude);

    /// Use GetIsOK() to verify successful initialisation. `device` = HIP device ordinal.
    Renderer(unsigned viewportWidth, unsigned viewportHeight, const Camera &camera, int device = 0);
    ~Renderer();
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    /// May reorder `primitives`; keeps nothing of them afterwards.
    void SetPrimitives(std::vector<Primitive *> &primitives, bool printInfo);
    bool UpdateViewportSize(unsigned width, unsigned height);
    bool SetCamera(const Camera &cam);

    void SetSunAzimuth(float azimuth) { Lighting.azimuth = azimuth; ResetPathTracing(); }
    float GetSunAzimuth() const { return Lighting.azimuth; }
    void SetSunAltitude(float altitude) { Lighting.altitude = altitude; ResetPathTracing(); }
    float GetSunAltitude() const { return Lighting.altitude; }
    void SetSunDirectLighting(bool enabled = true) { Lighting.directLightingEnabled = enabled; ResetPathTracing(); }
    bool IsSunDirectLightingEnabled() const { return Lighting.directLightingEnabled; }

    /// Use radius = 0 to effectively disable the user-controlled sphere.
    void SetUserSphere(const Vec3f &pos, float radius, float emittance);
    void SetUserSphereSpecular(bool specular) { SetFlag(SPECULAR, specular); }
    void SetUserSphereFuzzy(bool fuzzy) { SetFlag(FUZZY, fuzzy); }
    void SetUserSphereRadius(float radius) { UserSphere.radius = radius; ResetPathTracing(); }
    void SetUserSpherePos(const Vec3f &pos) { UserSphere.pos = pos; ResetPathTracing(); }
    void SetUserSphereEmittance(float em);
    Vec3f GetUserSpherePos() const { return UserSphere.pos; }
    float GetUserSphereRadius() const { return UserSphere.radius; }
    float GetUserSphereEmittance() const { return UserSphere.emittance; }

    void RenderDirectLighting();
    void RestartPathTracing(unsigned pathsPerPass, unsigned pathsPerPixel);
    /// Renders one progressive pass (if paths remain); returns paths per pixel rendered so far.
    unsigned RenderPathTracingPass();
    unsigned GetPathsPerPixel() const { return PathTracing.pathsPerPixel; }
    bool GetIsOK() const { return IsOK; }

    // ---- additions -------------------------------------------------------------------------
    /// RGBA32F, tile-sized, row 0 = bottom row. Synchronises.
    bool ReadDirectLighting(float *rgba);
    /// Accumulated radiance; normalized = divided by the paths rendered (what ptracingNormalize shows).
    bool ReadRadiance(float *rgba, bool normalized);
    bool Finish();
    void SetMaxPathSegments(unsigned n) { MaxPathSegments = n; ResetPathTracing(); }
    void SetMinWeight(float w) { MinWeight = w; ResetPathTracing(); }
    void SetSeed(uint32_t seed) { RndGen.seed(seed); ResetPathTracing(); }
    /// Opts in to the nearer-child-first BVH walk for trees of at least minPrims primitives (0xffffffff = never, the default): ~10 % faster,
    /// soak-verified but NOT proven to return the reference's winner — the reference's phantom hits of grazing triangles are a property
    /// of its own visiting order (include/gpuart_hip.h gpuart_hip_set_nearest_first). Restarts the accumulation like every setter.
    bool SetNearestFirst(uint32_t minPrims);
    /// Restricts this renderer to a tile of the frame (screen-space sharding across GPUs).
    bool SetTile(unsigned x0, unsigned y0, unsigned w, unsigned h);
    /// Row bands interleaved with other renderers (rank r of N: y0 = bandRows*r, bandStride = bandRows*N).
    bool SetInterleavedTile(unsigned x0, unsigned y0, unsigned w, unsigned localRows, unsigned bandRows, unsigned bandStride);
    /// One frame on several GPUs: this renderer renders share `rank` of `nranks` — 8-row bands of
    /// the frame dealt round-robin (gpuart_hip_share_of_rank). Every rank sets up the same scene, camera, lighting and
    /// seed, so all ranks draw the same RandSeed sequence; nothing is exchanged per pass.
    bool SetShare(int rank, int nranks);
    /// Assembles the shares of `ranks[0..n)` (renderer k = SetShare(k, n), one GPU each, all driven by this thread) into
    /// one frame in host memory: W*H RGBA32F, row 0 = bottom row. The rows travel over RCCL to `root`'s GPU
    /// (gpuart_hip_gather_all), normalized = divided by the paths rendered. Replaces the normalise-to-display step of the
    /// reference (src/renderer.cpp:601-616) for a frame that lives on several GPUs.
    static bool GatherRadiance(Renderer *const *ranks, int n, int root, bool normalized, float *fullFrame);
    /// Gives the communicator GatherRadiance made back (ncclCommDestroy per rank) as a named, watched phase of its own, instead
    /// of leaving it to the renderers' destructors at exit. False if a rank's destroy failed or did not return within its bound
    /// (gpuart_hip_comm_stuck() tells which): the caller should then end the process without unwinding.
    static bool ReleaseCommunicator(Renderer *const *ranks, int n);
    unsigned GetNumPathsRendered() const { return PathTracing.numPathsRendered; }
    /// Progressive-render checkpoint: accumulator + pass counters + RNG state of this tile.
    /// After LoadCheckpoint the following passes are bit-identical to those of the uninterrupted run. The
    /// scene, camera, lighting and viewport/tile must have been set up as they were when saving.
    bool SaveCheckpoint(const char *fileName);
    bool LoadCheckpoint(const char *fileName);
    /// Continues the current accumulation towards a new target without clearing it (RestartPathTracing would): used
    /// after LoadCheckpoint to render on to more paths per pixel than the checkpointed run asked for.
    void ExtendPathTracing(unsigned pathsPerPass, unsigned pathsPerPixel);
    gpuart_hip_ctx *GetBackend() const { return Backend; }
    /// What the last SetPrimitives spent, in ms: the whole call, the BVH build, its compilation, re-layout + upload (for tools/setprims_time.py).
    const double *GetLastSetPrimitivesMs() const { return LastSetPrimitivesMs; }
    unsigned GetTileWidth() const { return Tile.w; }
    unsigned GetTileHeight() const { return Tile.h; }
    const BoundingVolumesHierarchy &GetBVH() const { return Tree; }
    gpuart_params MakeParams() const;

private:
    enum UserSphereFlags : uint32_t { EM_NONZERO = 1u << 0, SPECULAR = 1u << 1, FUZZY = 1u << 2 };

    bool IsOK = false;
    double LastSetPrimitivesMs[4] = {0, 0, 0, 0};
    gpuart_hip_ctx *Backend = nullptr;
    BoundingVolumesHierarchy Tree;
    struct { unsigned width, height; } Viewport{0, 0};
    struct { unsigned x, y, w, h; } Tile{0, 0, 0, 0};  ///< the part of the frame this renderer owns
    Camera CurrentCamera;
    struct { float azimuth, altitude; bool directLightingEnabled; } Lighting;
    struct { Vec3f pos; float radius, emittance; uint32_t flags; } UserSphere;
    struct { unsigned numPathsRendered, pathsPerPixel, pathsPerPass; } PathTracing;
    unsigned MaxPathSegments = 5;  ///< MAX_PATH_SEGMENTS of the reference shader
    float MinWeight = 0.01f;       ///< MIN_WEIGHT of the reference shader
    std::mt19937 RndGen;           ///< default seed, never re-seeded by the reference

    void SetFlag(uint32_t flag, bool on);
    void ResetPathTracing();
    bool Check(int status, const char *what);
};

}  // namespace gpuart
#endif
