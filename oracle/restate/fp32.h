// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product.
//
// Strict-fp32 vocabulary for the CPU restatement of gpuart's GLSL device code, following the
// evaluation order that Mesa llvmpipe gives the reference shaders (re-verified
// against probes of the unmodified reference shaders by tests/golden/make_golden.py):
//   * no FMA contraction anywhere except inside sin()/cos()  (build with -ffp-contract=off);
//   * dot(vec3) = ((z*z' + y*y') + x*x')   (NIR lowers fdot as a reduction from the LAST
//     component); length = sqrt(dot); normalize(v) = v * (1/sqrt(dot(v,v)));
//   * '/', sqrt, 1/x are IEEE correctly rounded.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};

static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/// llvmpipe's dot(vec3,vec3): reduction starting from the last component.
static inline float dot3(V3 a, V3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
static inline V3 cross3(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float rsq(float x) { return 1.0f / sqrtf(x); }
static inline V3 normalize3(V3 v) { return v * rsq(dot3(v, v)); }
static inline float length3(V3 v) { return sqrtf(dot3(v, v)); }
static inline float mixf(float x, float y, float a) { return x + (y - x) * a; }

// ---- transcendental emulation of llvmpipe (gallivm) ------------------------------------------
// sin/cos: Cephes/sse_mathfun sin_ps-style with FMAs inside the polynomial stages.
void sincos_lp(float a, float *s, float *c);
static inline float sin_lp(float a) { float s, c; sincos_lp(a, &s, &c); return s; }
static inline float cos_lp(float a) { float s, c; sincos_lp(a, &s, &c); return c; }
/// pow(x,y) as gallivm computes it: exp2(y*log2(x)) with polynomial log2/exp2 approximations.
float pow_lp(float x, float y);

}  // namespace orc
