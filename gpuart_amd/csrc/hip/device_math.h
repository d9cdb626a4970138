// device_math.h — strict-fp32 device vocabulary of the gfx950 path tracer.
//
// The results must equal the reference GLSL as Mesa llvmpipe evaluates it (DESIGN.md "Numerics"):
// built with -ffp-contract=off, IEEE-correct '/' and sqrtf, denormals flushed
// (-fgpu-flush-denormals-to-zero, as llvmpipe runs with FTZ/DAZ); the only fused operations are
// the explicit fmaf() calls below (llvmpipe's sin/cos/pow and attribute interpolation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gd {

struct F3 {
    float x, y, z;
};

#define GD_FN __device__ __forceinline__

GD_FN F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
GD_FN F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
GD_FN F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
GD_FN F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
GD_FN F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
GD_FN F3 operator*(float s, F3 a) { return f3(a.x * s, a.y * s, a.z * s); }
GD_FN F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
GD_FN F3 xyz(float4 q) { return f3(q.x, q.y, q.z); }

/// GLSL dot(vec3,vec3) as llvmpipe lowers it: a reduction starting at the LAST component.
GD_FN float dot3(F3 a, F3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
GD_FN F3 cross3(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
GD_FN float rsq(float x) { return 1.0f / sqrtf(x); }
GD_FN F3 normalize3(F3 v) { return v * rsq(dot3(v, v)); }
GD_FN float length3(F3 v) { return sqrtf(dot3(v, v)); }
GD_FN float mixf(float x, float y, float a) { return x + (y - x) * a; }
GD_FN F3 reflect3(F3 I, F3 N) { return I - N * (2 * dot3(N, I)); }

// ---- hash RNG (reference shaders/noise.glsl:13-46): one Jenkins one-at-a-time round ----------
GD_FN uint32_t hash1(uint32_t x) {
    x += (x << 10);
    x ^= (x >> 6);
    x += (x << 3);
    x ^= (x >> 11);
    x += (x << 15);
    return x;
}
GD_FN float float_construct(uint32_t m) { return __uint_as_float((m & 0x007FFFFFu) | 0x3F800000u) - 1.0f; }
GD_FN float random1(float x) { return float_construct(hash1(__float_as_uint(x))); }
GD_FN float random2(float x, float y) { return float_construct(hash1(__float_as_uint(x) ^ hash1(__float_as_uint(y)))); }
GD_FN float random3(F3 v) {
    return float_construct(hash1(__float_as_uint(v.x) ^ hash1(__float_as_uint(v.y)) ^ hash1(__float_as_uint(v.z))));
}
GD_FN float random4(float4 v) {
    return float_construct(hash1(__float_as_uint(v.x) ^ hash1(__float_as_uint(v.y)) ^ hash1(__float_as_uint(v.z)) ^
                                 hash1(__float_as_uint(v.w))));
}

// ---- llvmpipe's sin/cos: Cephes-style, FMAs exactly where gallivm emits fmuladd ----------------
GD_FN void sincos_lp(float a, float &s_out, float &c_out) {
    const float FOPI = 1.27323954473516f;
    const float DP1 = -0.78515625f, DP2 = -2.4187564849853515625e-4f, DP3 = -3.77489497744594108e-8f;
    uint32_t sign_in = __float_as_uint(a) & 0x80000000u;
    float x = __uint_as_float(__float_as_uint(a) & 0x7fffffffu);
    float y = x * FOPI;
    int32_t j = (int32_t)y;
    j = (j + 1) & ~1;
    y = (float)j;
    x = fmaf(y, DP1, x);
    x = fmaf(y, DP2, x);
    x = fmaf(y, DP3, x);
    float z = x * x;
    float yc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    yc = fmaf(yc, z, 4.166664568298827e-2f);
    yc *= z;
    yc *= z;
    yc = fmaf(z, -0.5f, yc);
    yc += 1.0f;
    float ys = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ys = fmaf(ys, z, -1.6666654611e-1f);
    ys *= z;
    ys = fmaf(ys, x, x);
    float rs = (j & 2) ? yc : ys;
    s_out = __uint_as_float(__float_as_uint(rs) ^ (sign_in ^ ((uint32_t)(j & 4) << 29)));
    int32_t jc = j - 2;
    float rc = (jc & 2) ? yc : ys;
    c_out = __uint_as_float(__float_as_uint(rc) ^ (((uint32_t)(~jc & 4)) << 29));
}

// ---- llvmpipe's pow(x, y) = exp2(y * log2(x)), gallivm polynomials (even/odd Horner, fused) -----
GD_FN float log2_lp(float x) {
    const float P0 = 2.88539009343309178325f, P1 = 0.961791550404184197881f, P2 = 0.577440339438736392009f,
                P3 = 0.403343858251329912514f, P4 = 0.406718052498846252698f;
    uint32_t i = __float_as_uint(x);
    float logexp = (float)((int32_t)((i >> 23) & 0xff) - 127);
    float mant = __uint_as_float((i & 0x007fffffu) | 0x3f800000u);
    float y = (mant - 1.0f) / (mant + 1.0f);
    float z = y * y;
    float z2 = z * z;
    float even = fmaf(z2, fmaf(z2, P4, P2), P0);
    float odd = fmaf(z2, P3, P1);
    float p = fmaf(odd, z, even);
    return fmaf(y, p, logexp);
}
GD_FN float exp2_lp(float x) {
    const float P0 = 1.0f, P1 = 0.693153073200168932794f, P2 = 0.240153617044375388211f,
                P3 = 0.0558263180532956664775f, P4 = 0.00898934009049466391101f, P5 = 0.00187757667519147912699f;
    if (x > 128.0f) x = 128.0f;  // gallivm clamps the argument to [-126.99999, 128]: 2^128 is +inf ((128 + 127) << 23)
    if (x < -126.99999f) x = -126.99999f;
    float ip = floorf(x);
    float fp = x - ip;
    float e = __uint_as_float((uint32_t)(((int32_t)ip + 127) << 23));
    float f2 = fp * fp;
    float even = fmaf(f2, fmaf(f2, P4, P2), P0);
    float odd = fmaf(f2, fmaf(f2, P5, P3), P1);
    float p = fmaf(odd, fp, even);
    return e * p;
}
/// Special bases as llvmpipe answers them (probed; tests/golden/sky_wild.npz): NaN -> 0, a negative base (not zero; -inf too) ->
/// NaN, +inf and every base whose power overflows -> +inf, +-0 and denormals -> 0 (the last three fall out of the polynomials).
GD_FN float pow_lp(float x, float y) {
    const uint32_t b = __float_as_uint(x);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0.0f;
    if ((b >> 31) && (b & 0x7f800000u)) return __uint_as_float(0x7fc00000u);
    return exp2_lp(log2_lp(x) * y);
}

}  // namespace gd
