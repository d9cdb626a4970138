// core.cpp — primitive bounding boxes and their serialisation into compiled-BVH quads.
// Follows the reference's formats exactly (src/core.cpp:36-245): the bytes are the interchange
// format consumed by gpuart_hip_upload_bvh.
#include "core.h"

#include <algorithm>
#include <cstring>
#include <ostream>

namespace gpuart {

namespace {
inline void put(Primitive::Data &d, std::initializer_list<float> v) { d.insert(d.end(), v); }
inline float as_float(uint32_t u) { float f; std::memcpy(&f, &u, sizeof f); return f; }
}  // namespace

void Primitive::StoreIntoBVH(Data &data) const {
    put(data, {as_float((uint32_t)GetType()), RGBA_PAD, RGBA_PAD, RGBA_PAD});
    StoreDataIntoBVH(data);
}

// ---- sphere: box = centre -/+ radius; payload {c, r} (reference src/core.cpp:36-65) ------------
Sphere::Sphere(const Vec3f &center, float radius) : Center(center), Radius(radius) {
    SetBox(Vec3f(center.x - radius, center.y - radius, center.z - radius),
           Vec3f(center.x + radius, center.y + radius, center.z + radius));
}
void Sphere::StoreDataIntoBVH(Data &d) const { put(d, {Center.x, Center.y, Center.z, Radius}); }
void Sphere::PrintBVH(Data::const_iterator &it, std::ostream &os) {
    os << "{ (" << it[0] << ", " << it[1] << ", " << it[2] << "), " << it[3] << " }";
    it += 4;
}

// ---- disc: box of the sphere with the same radius; payload {c, r}{n, pad} (src/core.cpp:80-115) --
Disc::Disc(const Vec3f &center, const Vec3f &normal, float radius) : Center(center), Normal(normal), Radius(radius) {
    SetBox(Vec3f(center.x - radius, center.y - radius, center.z - radius),
           Vec3f(center.x + radius, center.y + radius, center.z + radius));
}
void Disc::StoreDataIntoBVH(Data &d) const {
    put(d, {Center.x, Center.y, Center.z, Radius, Normal.x, Normal.y, Normal.z, RGBA_PAD});
}
void Disc::PrintBVH(Data::const_iterator &it, std::ostream &os) {
    os << "{ (" << it[0] << ", " << it[1] << ", " << it[2] << "), " << it[3] << ", (" << it[4] << ", " << it[5] << ", "
       << it[6] << ") }";
    it += 8;
}

// ---- triangle: payload {v0,pad}{v1,pad}{v2,pad} (src/core.h:119-141, src/core.cpp:136-168) ------
Triangle::Triangle(const Vec3f &v0, const Vec3f &v1, const Vec3f &v2) {
    Vert[0] = v0; Vert[1] = v1; Vert[2] = v2;
    CalcBoundingBox();
}
void Triangle::CalcBoundingBox() {
    Vec3f lo(9e+19f, 9e+19f, 9e+19f), hi(-9e+19f, -9e+19f, -9e+19f);
    for (const Vec3f &v : Vert) {
        if (v.x < lo.x) lo.x = v.x;
        if (v.x > hi.x) hi.x = v.x;
        if (v.y < lo.y) lo.y = v.y;
        if (v.y > hi.y) hi.y = v.y;
        if (v.z < lo.z) lo.z = v.z;
        if (v.z > hi.z) hi.z = v.z;
    }
    SetBox(lo, hi);
}
void Triangle::StoreDataIntoBVH(Data &d) const {
    for (const Vec3f &v : Vert) put(d, {v.x, v.y, v.z, RGBA_PAD});
}
void Triangle::PrintBVH(Data::const_iterator &it, std::ostream &os) {
    os << "{ ";
    for (int i = 0; i < 3; i++, it += 4) os << "(" << it[0] << ", " << it[1] << ", " << it[2] << ")" << (i < 2 ? ", " : "");
    os << " }";
}

// ---- cone: derived constants in double precision, then narrowed (src/core.cpp:191-245) ---------
Cone::Cone(const Vec3f &center1, const Vec3f &center2, float radius1, float radius2)
    : Center1(center1), Center2(center2), Radius1(radius1), Radius2(radius2) {
    const Vec3d c1(center1), c2(center2);
    AxisLen = (float)(c2 - c1).length();
    const Vec3d axis = (c2 - c1) / AxisLen;
    UnitAxis = axis;
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
    DotAxC1 = (float)(axis * c1);
    // box of the frustum with hemispherical caps (slightly larger than necessary)
    SetBox(Vec3f(std::min(center1.x - radius1, center2.x - radius2), std::min(center1.y - radius1, center2.y - radius2),
                 std::min(center1.z - radius1, center2.z - radius2)),
           Vec3f(std::max(center1.x + radius1, center2.x + radius2), std::max(center1.y + radius1, center2.y + radius2),
                 std::max(center1.z + radius1, center2.z + radius2)));
}
void Cone::StoreDataIntoBVH(Data &d) const {
    put(d, {Center1.x, Center1.y, Center1.z, Radius1, Center2.x, Center2.y, Center2.z, Radius2, UnitAxis.x, UnitAxis.y,
            UnitAxis.z, AxisLen, WidthCoeff, CosB, DotAxC1, RGBA_PAD});
}
/// Decodes the 16 floats StoreDataIntoBVH writes. (The reference's Cone::PrintBVH reads a stale
/// 20-float layout, src/core.cpp:248-281; the stored format is what the device consumes.)
void Cone::PrintBVH(Data::const_iterator &it, std::ostream &os) {
    os << "{ (" << it[0] << ", " << it[1] << ", " << it[2] << "), " << it[3] << ", (" << it[4] << ", " << it[5] << ", "
       << it[6] << "), " << it[7] << ", axis (" << it[8] << ", " << it[9] << ", " << it[10] << "), len " << it[11]
       << ", widthCoeff " << it[12] << ", cosB " << it[13] << ", dotAxC1 " << it[14] << " }";
    it += 16;
}

}  // namespace gpuart
