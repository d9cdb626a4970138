// core.h — scene primitives of the Renderer/Scene API (reference src/core.h:40-204).
// Same public surface (Primitive + Sphere/Disc/Triangle/Cone, Primitive::Data, StoreIntoBVH,
// GetXmin..GetZmax, PrintBVH); GL-free: Data is std::vector<float>.
#ifndef GPUART_CORE_H
#define GPUART_CORE_H

#include <cstdint>
#include <iosfwd>
#include <vector>

#include "math_types.h"

#define RGBA_PAD 0.0f
#define RGBA_ELEMS 4

namespace gpuart {

/// Values are shared with the device code (the type word of each compiled primitive).
enum Primitive_t { SPHERE = 0, DISC = 1, TRIANGLE = 2, CONE = 3 };

class Primitive {
public:
    /// A compiled scene: RGBA32F "quads" (the reference's GL buffer-texture contents).
    typedef std::vector<float> Data;

    virtual ~Primitive() {}

    /// Appends {type bits, pad, pad, pad} followed by the type's payload quads.
    void StoreIntoBVH(Data &data) const;
    /// Number of floats StoreIntoBVH appends: the type quad + 1 (sphere), 2 (disc), 3 (triangle) or 4 (cone) data quads.
    size_t GetBVHDataLength() const {
        static const size_t DATA_QUADS[4] = {1, 2, 3, 4};
        return 4 * (1 + DATA_QUADS[GetType() & 3]);
    }

    float GetXmin() const { return Xmin; }
    float GetXmax() const { return Xmax; }
    float GetYmin() const { return Ymin; }
    float GetYmax() const { return Ymax; }
    float GetZmin() const { return Zmin; }
    float GetZmax() const { return Zmax; }

protected:
    /// World-space bounding box, set by the derived constructors.
    float Xmin, Xmax, Ymin, Ymax, Zmin, Zmax;
    void SetBox(const Vec3f &lo, const Vec3f &hi) {
        Xmin = lo.x; Ymin = lo.y; Zmin = lo.z;
        Xmax = hi.x; Ymax = hi.y; Zmax = hi.z;
    }

private:
    virtual Primitive_t GetType() const = 0;
    virtual void StoreDataIntoBVH(Data &data) const = 0;
};

class Sphere : public Primitive {
    Vec3f Center;
    float Radius;
    void StoreDataIntoBVH(Data &data) const override;
    Primitive_t GetType() const override { return SPHERE; }

public:
    Sphere() : Sphere(Vec3f(0, 0, 0), 1) {}
    Sphere(const Vec3f &center, float radius);
    static void PrintBVH(Data::const_iterator &it, std::ostream &os);
};

class Disc : public Primitive {
    Vec3f Center, Normal;
    float Radius;
    void StoreDataIntoBVH(Data &data) const override;
    Primitive_t GetType() const override { return DISC; }

public:
    Disc() : Disc(Vec3f(0, 0, 0), Vec3f(0, 0, 1), 1) {}
    Disc(const Vec3f &center, const Vec3f &normal, float radius);
    static void PrintBVH(Data::const_iterator &it, std::ostream &os);
};

class Triangle : public Primitive {
    Vec3f Vert[3];
    void CalcBoundingBox();
    void StoreDataIntoBVH(Data &data) const override;
    Primitive_t GetType() const override { return TRIANGLE; }

public:
    Triangle() : Triangle(Vec3f(0, 0, 0), Vec3f(1, 0, 0), Vec3f(1, 1, 0)) {}
    Triangle(float v0x, float v0y, float v0z, float v1x, float v1y, float v1z, float v2x, float v2y, float v2z)
        : Triangle(Vec3f(v0x, v0y, v0z), Vec3f(v1x, v1y, v1z), Vec3f(v2x, v2y, v2z)) {}
    Triangle(const Vec3f &v0, const Vec3f &v1, const Vec3f &v2);
    static void PrintBVH(Data::const_iterator &it, std::ostream &os);
};

/// Conical frustum between two centres with two radii.
class Cone : public Primitive {
    Vec3f Center1, Center2;
    float Radius1, Radius2;
    Vec3f UnitAxis;    ///< unit (Center2 - Center1)
    float AxisLen;     ///< |Center2 - Center1|
    float WidthCoeff;  ///< (Radius2 - Radius1) / AxisLen
    float CosB;        ///< cosine of the base angle
    float DotAxC1;     ///< UnitAxis . Center1
    void StoreDataIntoBVH(Data &data) const override;
    Primitive_t GetType() const override { return CONE; }

public:
    Cone(const Vec3f &center1, const Vec3f &center2, float radius1, float radius2);
    static void PrintBVH(Data::const_iterator &it, std::ostream &os);
};

}  // namespace gpuart
#endif
