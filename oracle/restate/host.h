// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product.
//
// CPU restatement of the reference's host-side arithmetic on the hot path: primitive
// serialisation + world AABBs (src/core.cpp), BVH build + compile (src/bvh.cpp), camera
// screen basis (src/renderer.cpp:135-166), Sun direction (src/renderer.h:175-179), per-pass
// RandSeed (src/renderer.cpp:585-589) and PixelSize (src/renderer.cpp:573-574).
//
// C++ (not C) on purpose: the reference's results depend on libstdc++'s std::sort (unstable
// order of equal keys), std::mt19937 and std::uniform_real_distribution<float>; restating
// through the same library is the only way to reproduce them.
#pragma once
#include <cstdint>
#include <random>
#include <vector>

namespace orc {

/// A scene primitive in "description" form: type + up to 8 floats.
///   sphere  : cx cy cz r
///   disc    : cx cy cz  nx ny nz  r
///   triangle: v0 v1 v2 (9 floats)
///   cone    : c1 (3) c2 (3) r1 r2
struct PrimDesc {
    int type;
    float f[9];
};

struct PrimRec {
    int type;
    float bbmin[3], bbmax[3];
    float data[16];  ///< StoreDataIntoBVH payload (4,8,12,16 floats)
    int ndata;
};

PrimRec MakePrim(const PrimDesc &d);

/// Build (maxNumLevels, minPrimitivesPerNode as in src/renderer.cpp:454: 1024, 2) + Compile.
/// Returns the flat RGBA32F quad array; `order` receives the primitive permutation.
std::vector<float> BuildAndCompileBVH(std::vector<PrimRec> prims, unsigned maxNumLevels, unsigned minPrimsPerNode,
                                      int *maxDepth);

struct CameraBasis {
    float pos[3], bottomLeft[3], deltaHorz[3], deltaVert[3];
};
CameraBasis CameraScreenBasis(const float pos[3], const float dir[3], const float up[3], float fovY, float screenDist,
                              unsigned W, unsigned H);
float PixelSize(float fovY, float screenDist, unsigned H);
void SunDirection(float azimuth, float altitude, float out[3]);

struct RandSeedGen {
    std::mt19937 gen;  ///< default-constructed (seed 5489), never re-seeded (src/renderer.h:165)
    void next(float out[4]) {
        std::uniform_real_distribution<float> d(0, 1);
        for (int i = 0; i < 4; i++) out[i] = d(gen);
    }
};

}  // namespace orc
