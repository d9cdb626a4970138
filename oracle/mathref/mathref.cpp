// oracle/mathref/mathref.cpp — TEST INFRASTRUCTURE, container only.
//
// The one C++ file of the reference that compiles without the un-vendored nanogui is src/math_types.h (it includes only
// <math.h>, <iostream>, <initializer_list>). This wrapper includes it FROM WHERE IT LIES (-I/root/reference/src, see
// oracle/Makefile: target mathref -> oracle/_ref/libmathref.so; nothing of it is copied) and exposes the reference's own
// Vec3<float> / Vec3<double> arithmetic through a C ABI, so that the host-side restatements (oracle/restate/host.cpp and
// gpuart_amd/csrc/host/) are pinned against the reference's compiled code for everything that is Vec3 arithmetic:
//   * every Vec3 operation on its own (mref_vec3f_ops / mref_vec3d_ops);
//   * the expression of Renderer::GetSunDirection, src/renderer.h:175-179, evaluated on the reference's Vec3f;
//   * the statements of Cone::Cone, src/core.cpp:191-226 (constants in double, world box), in the payload order of :230-245;
//   * the statements of Renderer::SetCamera that compute the screen basis, src/renderer.cpp:139-149 and :158-161, and of
//     the PixelSize uniform, src/renderer.cpp:573-574, evaluated on the reference's Vec3f with its PI (src/renderer.cpp:48).
// Renderer itself (renderer.cpp) cannot be compiled here — it includes <nanogui/nanogui.h> — so the last two are the
// reference's statements re-typed around the reference's class, not the reference's functions; the Vec3 methods they call
// (normalized, ^, *, length, vroty, vrotz with their float cos/sin/sqrt overloads and the 1/a multiply of operator/) ARE
// the reference's compiled code. tests/golden/make_host_golden.py turns this library into tests/golden/host_math.npz.
#include "math_types.h"

#include <algorithm>
#include <cmath>

using gpuart::Vec3d;
using gpuart::Vec3f;

#define MREF_PI 3.1415926f  // src/renderer.cpp:48

extern "C" {

/// out[0..2] = Vec3f(1,0,0).vroty(-altitude).vrotz(azimuth)   (src/renderer.h:175-179)
void mref_sun_direction(float azimuth, float altitude, float out[3]) {
    Vec3f v = Vec3f(1, 0, 0).vroty(-altitude).vrotz(azimuth);
    v.storeIn(out);
}

/// out = {bottomLeft, deltaHorz, deltaVert} (9 floats), out[9] = PixelSize   (src/renderer.cpp:139-149,158-161,573-574)
void mref_camera_basis(const float pos[3], const float dir[3], const float upv[3], float fovY, float screenDist, unsigned width,
                       unsigned height, float out[10]) {
    const Vec3f Pos(pos), Dir(dir), Up(upv);
    float aspect = (float)width / height;
    Vec3f up = ((Dir ^ Up) ^ Dir).normalized();
    Vec3f target = Pos + Dir.normalized() * screenDist;
    Vec3f a = (Dir.normalized() ^ up) * screenDist * aspect * std::tan(fovY / 2 * MREF_PI / 180);
    Vec3f b = up * a.length() / aspect;
    Vec3f bl = target - a - b, dh = 2 * a, dv = 2 * b;
    bl.storeIn(out); dh.storeIn(out + 3); dv.storeIn(out + 6);
    out[9] = 2 * screenDist * std::tan(fovY / 2 * MREF_PI / 180) / height;
}

/// The statements of Cone::Cone (src/core.cpp:191-226: derived constants computed in double, world box as if the cone had
/// hemispherical caps) and the payload order of Cone::StoreDataIntoBVH (:230-245), on the reference's Vec3f / Vec3d.
/// out[0..15] = {c1, r1}{c2, r2}{unitAxis, axisLen}{widthCoeff, cosB, dotAxC1, 0}, out[16..21] = {xmin, ymin, zmin, xmax, ymax, zmax}
void mref_cone(const float c1[3], const float c2[3], float Radius1, float Radius2, float out[22]) {
    const Vec3f center1(c1), center2(c2);
    float AxisLen, WidthCoeff, CosB, DotAxC1;
    Vec3f UnitAxis;
    Vec3d vc1(center1), vc2(center2);
    AxisLen = (float)(vc2 - vc1).length();
    Vec3d vd = (vc2 - vc1) / AxisLen;
    UnitAxis = vd;
    WidthCoeff = (Radius2 - Radius1) / AxisLen;
    if (fabs(Radius1 - Radius2) < 1.0e-7)
        CosB = 0.0f;
    else if (Radius1 > Radius2) {
        float h = Radius1 * AxisLen / (Radius1 - Radius2);
        CosB = (float)(Radius1 / sqrt((double)h * h + (double)Radius1 * Radius1));
    } else {
        float h = Radius2 * AxisLen / (Radius2 - Radius1);
        CosB = (float)(-Radius2 / sqrt((double)h * h + (double)Radius2 * Radius2));
    }
    DotAxC1 = (float)(vd * vc1);
    out[0] = center1.x; out[1] = center1.y; out[2] = center1.z; out[3] = Radius1;
    out[4] = center2.x; out[5] = center2.y; out[6] = center2.z; out[7] = Radius2;
    out[8] = UnitAxis.x; out[9] = UnitAxis.y; out[10] = UnitAxis.z; out[11] = AxisLen;
    out[12] = WidthCoeff; out[13] = CosB; out[14] = DotAxC1; out[15] = 0.0f;
    out[16] = std::min(center1.x - Radius1, center2.x - Radius2); out[19] = std::max(center1.x + Radius1, center2.x + Radius2);
    out[17] = std::min(center1.y - Radius1, center2.y - Radius2); out[20] = std::max(center1.y + Radius1, center2.y + Radius2);
    out[18] = std::min(center1.z - Radius1, center2.z - Radius2); out[21] = std::max(center1.z + Radius1, center2.z + Radius2);
}

/// Every Vec3f operation once: out = {length, sqrlength, dot} + normalized(a) + a^b + a+b + a-b + a*s + s*a + a/s +
/// a.vrotx(s) + a.vroty(s) + a.vrotz(s) + (-a)   (3 + 11*3 = 36 floats)
void mref_vec3f_ops(const float av[3], const float bv[3], float s, float out[36]) {
    Vec3f a(av), b(bv);
    out[0] = a.length(); out[1] = a.sqrlength(); out[2] = a * b;
    Vec3f r[11] = {a.normalized(), a ^ b, a + b, a - b, a * s, s * a, a / s, a.vrotx(s), a.vroty(s), a.vrotz(s), -a};
    for (int i = 0; i < 11; i++) r[i].storeIn(out + 3 + 3 * i);
}

/// The same on Vec3d (Cone::Cone's constants are computed in double, src/core.cpp:191-226)
void mref_vec3d_ops(const double av[3], const double bv[3], double s, double out[36]) {
    Vec3d a(av), b(bv);
    out[0] = a.length(); out[1] = a.sqrlength(); out[2] = a * b;
    Vec3d r[11] = {a.normalized(), a ^ b, a + b, a - b, a * s, s * a, a / s, a.vrotx(s), a.vroty(s), a.vrotz(s), -a};
    for (int i = 0; i < 11; i++) { out[3 + 3 * i] = r[i].x; out[4 + 3 * i] = r[i].y; out[5 + 3 * i] = r[i].z; }
}

}  // extern "C"
