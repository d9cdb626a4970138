// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product.
//
// CPU restatement of gpuart's GLSL device code (reference shaders/*.glsl), one function per
// reference function, each citing the lines it follows. Operates on the reference's canonical
// compiled-BVH layout (RGBA32F "quads", src/bvh.cpp:161-222).
#pragma once
#include <cstddef>
#include <cstdint>

#include "fp32.h"

namespace orc {

// Primitive types (src/core.h:40-46)
enum { P_SPHERE = 0, P_DISC = 1, P_TRIANGLE = 2, P_CONE = 3 };
// Node flags (src/bvh.h:48-52, shaders/bvh_intersection.glsl:40-44)
static const uint32_t BVH_LEAF = 1u << 31, BVH_IS_LOWER = 1u << 30, BVH_IS_ROOT = 1u << 29;
static const uint32_t BVH_FLAGS_MASK = BVH_LEAF | BVH_IS_LOWER | BVH_IS_ROOT;
// User-sphere flags (src/renderer.h:141-146)
static const uint32_t USPH_EM_NONZERO = 1u, USPH_SPECULAR = 2u, USPH_FUZZY = 4u;

struct Hit {
    float pos;   ///< ray parameter, <0 = miss
    V3 p;        ///< intersection point
    V3 n;        ///< unit normal facing the ray origin
    int ptype;   ///< primitive type, -1 = miss
};

/// Traversal statistics used for the algorithmic-bytes roofline.
struct TravStats {
    uint64_t rays = 0;        ///< closest-hit queries
    uint64_t iterations = 0;  ///< do-while iterations of the reference loop
    uint64_t nodes = 0;       ///< DISTINCT nodes whose box was tested (first visits only)
    uint64_t prim_tests[4] = {0, 0, 0, 0};  ///< tested primitives by type
    void add(const TravStats &o) {
        rays += o.rays; iterations += o.iterations; nodes += o.nodes;
        for (int i = 0; i < 4; i++) prim_tests[i] += o.prim_tests[i];
    }
    /// 48 B per distinct node + (16 + 16*len) per tested primitive, len = 1,2,3,4.
    uint64_t algorithmic_bytes() const {
        return 48 * nodes + 32 * prim_tests[0] + 48 * prim_tests[1] + 64 * prim_tests[2] + 80 * prim_tests[3];
    }
};

/// Uniforms of the directLighting / pathTracing programs (src/renderer.cpp:51-79).
struct Params {
    float sunDirAlt[4];
    int sunEnabled;
    float userSphere[4];
    float userSphereEm[3];
    uint32_t userSphereFlags;
    float pixelSize;
    float cameraPos[3];
    int maxSegments;     ///< MAX_PATH_SEGMENTS (shaders/path_tracing.glsl:138), reference value 5
    float minWeight;     ///< MIN_WEIGHT (shaders/path_tracing.glsl:139), reference value 0.01
};

// noise.glsl:13-46
uint32_t hash1(uint32_t x);
float random1(float x);
float random2(float x, float y);
float random3(V3 v);
float random4(V4 v);

// common.glsl:38-106
V3 GetOrthogonal(V3 v);
V3 GetRandomHemisphereDirection(V3 v, V3 randInput);
V3 rotate(V3 v, V3 axis, float sine, float cosine);
V3 GetRandomDirectionInsideCone(V3 v, V3 normal, float halfAngle, V3 randInput);

// sphere.glsl:31-70, disc.glsl:30-72, triangle.glsl:33-82, cone.glsl:30-135
void SphereIntersection(V3 rs, V3 rd, V3 c, float r, float &pos, V3 &p, V3 &n);
void DiscIntersection(V3 rs, V3 rd, V3 c, float r, V3 dn, float &pos, V3 &p, V3 &n);
void TriangleIntersection(V3 rs, V3 rd, V3 v0, V3 v1, V3 v2, float &pos, V3 &p, V3 &n);
void ConeIntersection(V3 rs, V3 rd, V4 cr1, V4 cr2, V4 axL, float widthCoeff, float cosB, float dotAxC1,
                      float &pos, V3 &p, V3 &n);

// bvh_intersection.glsl:229-354
bool IntersectsAABB(V3 rs, V3 rd, V3 rdiv, const float *tree, int addr, float &pos);
// bvh_intersection.glsl:360-457
void CheckBVHIntersection(V3 rs, V3 rd, const float *tree, Hit &h, TravStats *st);
// intersection.glsl:71-111
void CheckIntersectionInclUserSphere(V3 rs, V3 rd, const float *tree, const float userSphere[4], Hit &h,
                                     bool &userSphereHit, TravStats *st);
// sky.glsl:34-60
V3 GetSkyColor(V3 dir, const float sunDirAlt[4]);

// vertex.glsl:29-37 as rasterised by llvmpipe: plane-equation coefficients of the two fan triangles
// (coef = A.u, A.v, B.u, B.v; 3 floats each) and the per-pixel UV. Row 0 = bottom.
void QuadUVCoefs(int W, int H, float coef[12]);
void PixelUV(int x, int y, int W, int H, const float coef[12], float &u, float &v);
// cam_init.glsl:45-50
void CamInitPixel(int x, int y, int W, int H, const float pos[3], const float bl[3], const float dh[3],
                  const float dv[3], V3 &rstart, V3 &rdir);
// direct_lighting.glsl:134-207
V3 DirectLightingPixel(V3 rstart, V3 rdir, const float *tree, const Params &P, TravStats *st);
// path_tracing.glsl:133-256 (returns the sum over npaths paths, to be added to PrevRadiance)
void FirstSegmentRay(V3 rstart0, V3 rdir0, const Params &P, const float randSeed[4], int j, V3 &rstart, V3 &rdir);
V3 PathTracingPixel(V3 rstart0, V3 rdir0, const float *tree, const Params &P, const float randSeed[4],
                    int npaths, TravStats *st, uint64_t *npaths_segments);

}  // namespace orc
