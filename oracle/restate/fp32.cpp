// ORACLE — TEST INFRASTRUCTURE ONLY. llvmpipe transcendental emulation (see fp32.h).
#include "fp32.h"

namespace orc {

// gallivm lp_build_sin_or_cos (sse_mathfun lineage). The FMAs below are where Mesa emits
// llvm.fmuladd and the host CPU has FMA3 (the container and the GPU box are x86-64 with FMA).
void sincos_lp(float a, float *s_out, float *c_out) {
    const float FOPI = 1.27323954473516f;
    const float DP1 = -0.78515625f, DP2 = -2.4187564849853515625e-4f, DP3 = -3.77489497744594108e-8f;
    uint32_t sign_in = f2u(a) & 0x80000000u;
    float x = u2f(f2u(a) & 0x7fffffffu);
    float y = x * FOPI;
    int32_t j = (int32_t)y;  // truncation
    j = (j + 1) & ~1;
    y = (float)j;
    x = fmaf(y, DP1, x);
    x = fmaf(y, DP2, x);
    x = fmaf(y, DP3, x);
    float z = x * x;
    // cosine polynomial
    float yc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    yc = fmaf(yc, z, 4.166664568298827e-2f);
    yc *= z;
    yc *= z;
    yc = fmaf(z, -0.5f, yc);
    yc += 1.0f;
    // sine polynomial
    float ys = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ys = fmaf(ys, z, -1.6666654611e-1f);
    ys *= z;
    ys = fmaf(ys, x, x);
    // sin: poly select by (j&2), sign = input sign ^ (j&4)
    {
        float r = (j & 2) ? yc : ys;
        uint32_t sg = sign_in ^ ((uint32_t)(j & 4) << 29);
        *s_out = u2f(f2u(r) ^ sg);
    }
    // cos: uses j-2
    {
        int32_t jc = j - 2;
        float r = (jc & 2) ? yc : ys;
        uint32_t sg = ((uint32_t)(~jc & 4)) << 29;
        *c_out = u2f(f2u(r) ^ sg);
    }
}

// gallivm lp_build_pow = exp2(log2(x) * y); polynomial forms restated from the published Mesa
// algorithm (src/gallium/auxiliary/gallivm/lp_bld_arit.c, Mesa 23.2): log2 via
// y=(m-1)/(m+1), minimax polynomial in y^2; exp2 via floor split and a degree-5 polynomial.
// Evaluated Horner-style with even/odd split and fused multiply-adds (lp_build_polynomial).
static float poly_lp(float x, const float *c, int n) {
    float x2 = x * x;
    float even = 0, odd = 0;
    bool has_even = false, has_odd = false;
    for (int i = n; i--;) {
        if ((i & 1) == 0) {
            even = has_even ? fmaf(x2, even, c[i]) : c[i];
            has_even = true;
        } else {
            odd = has_odd ? fmaf(x2, odd, c[i]) : c[i];
            has_odd = true;
        }
    }
    if (has_odd) return fmaf(odd, x, even);
    return even;
}

static float log2_lp(float x) {
    static const float P[] = {2.88539009343309178325f, 0.961791550404184197881f, 0.577440339438736392009f,
                              0.403343858251329912514f, 0.406718052498846252698f};
    uint32_t i = f2u(x);
    int32_t e = (int32_t)((i >> 23) & 0xff) - 127;
    float logexp = (float)e;
    float mant = u2f((i & 0x007fffffu) | 0x3f800000u);
    float y = (mant - 1.0f) / (mant + 1.0f);
    float z = y * y;
    float p = poly_lp(z, P, 5);
    return fmaf(y, p, logexp);
}

static float exp2_lp(float x) {
    static const float P[] = {1.000000000000000000000f, 0.693153073200168932794f, 0.240153617044375388211f,
                              0.0558263180532956664775f, 0.00898934009049466391101f, 0.00187757667519147912699f};
    if (x > 128.0f) x = 128.0f;  // gallivm clamps the argument to [-126.99999, 128]: 2^128 is +inf ((128 + 127) << 23), what pow() of an infinite base gives
    if (x < -126.99999f) x = -126.99999f;
    float ip = floorf(x);
    float fp = x - ip;
    float e = u2f((uint32_t)(((int32_t)ip + 127) << 23));
    float p = poly_lp(fp, P, 6);
    return e * p;
}

// Special bases as llvmpipe answers them (probed: tests/golden/make_golden.py shade_wild and the table in oracle/README.md):
// NaN -> 0, a negative base (not zero; -inf too) -> NaN, +inf and every base whose power overflows -> +inf,
// +-0 and denormals -> 0. The last three fall out of the polynomials and the clamp; the first two are gallivm's edge handling.
float pow_lp(float x, float y) {
    const uint32_t b = f2u(x);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0.0f;              // NaN of either sign (0 * inf is the negative default NaN on x86)
    if ((b >> 31) && (b & 0x7f800000u)) return u2f(0x7fc00000u);  // (negative denormals are flushed to -0 first: log2 = -inf, result 0)
    return exp2_lp(log2_lp(x) * y);
}

}  // namespace orc
