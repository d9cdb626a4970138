// math_types.h — gpuart::Vec3<T>, the small vector type of the Renderer/Scene API.
// API-compatible with the reference's src/math_types.h (`*` between vectors is the dot product,
// `^` the cross product; vrotx/y/z rotate about the coordinate axes). Evaluation order of every
// float expression follows the reference (it feeds the camera basis and the Sun direction, which
// must be reproduced bit-for-bit): length = sqrt((x*x + y*y) + z*z), v/a = v * (1/a).
#ifndef GPUART_MATH_TYPES_H
#define GPUART_MATH_TYPES_H

#include <math.h>

#include <initializer_list>
#include <ostream>

namespace gpuart {

template <typename T>
class Vec3 {
public:
    T x{0}, y{0}, z{0};

    Vec3() = default;
    Vec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    explicit Vec3(const float a[3]) : x(a[0]), y(a[1]), z(a[2]) {}
    explicit Vec3(const double a[3]) : x(a[0]), y(a[1]), z(a[2]) {}
    Vec3(std::initializer_list<T> l) { *this = l; }
    template <typename U>
    Vec3(const Vec3<U> &o) : x(T(o.x)), y(T(o.y)), z(T(o.z)) {}

    Vec3 &operator=(std::initializer_list<T> l) {
        auto it = l.begin();
        x = *it++; y = *it++; z = *it;
        return *this;
    }
    void storeIn(float out[3]) const { out[0] = float(x); out[1] = float(y); out[2] = float(z); }

    T sqrlength() const { return x * x + y * y + z * z; }
    T length() const { return sqrt(sqrlength()); }
    Vec3 normalized() const { return *this / length(); }

    Vec3 &operator+=(const Vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    Vec3 &operator-=(const Vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    Vec3 &operator*=(T s) { x *= s; y *= s; z *= s; return *this; }
    Vec3 &operator/=(T s) { const T inv = 1 / s; return *this *= inv; }
    Vec3 &operator^=(const Vec3 &o) { return *this = *this ^ o; }
    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    const Vec3 &operator+() const { return *this; }
    bool operator==(const Vec3 &o) const { return x == o.x && y == o.y && z == o.z; }
    bool operator!=(const Vec3 &o) const { return !(*this == o); }

    friend Vec3 operator+(Vec3 a, const Vec3 &b) { return a += b; }
    friend Vec3 operator-(Vec3 a, const Vec3 &b) { return a -= b; }
    friend Vec3 operator*(Vec3 a, T s) { return a *= s; }
    friend Vec3 operator*(T s, Vec3 a) { return a *= s; }
    friend Vec3 operator/(Vec3 a, T s) { return a /= s; }
    /// dot product
    friend T operator*(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
    /// cross product
    friend Vec3 operator^(const Vec3 &a, const Vec3 &b) {
        return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    }

    // Rotations about the coordinate axes (right-handed), by angle or by (sine, cosine).
    Vec3 vrotx(T sine, T cosine) const { return Vec3(x, y * cosine - z * sine, y * sine + z * cosine); }
    Vec3 vroty(T sine, T cosine) const { return Vec3(z * sine + x * cosine, y, z * cosine - x * sine); }
    Vec3 vrotz(T sine, T cosine) const { return Vec3(x * cosine - y * sine, x * sine + y * cosine, z); }
    Vec3 vrotx(T angle) const { T c = cos(angle), s = sin(angle); return vrotx(s, c); }
    Vec3 vroty(T angle) const { T c = cos(angle), s = sin(angle); return vroty(s, c); }
    Vec3 vrotz(T angle) const { T c = cos(angle), s = sin(angle); return vrotz(s, c); }

    /// v rotated about the unit vector `a` (Rodrigues matrix form, as the reference writes it).
    static Vec3 rotate(const Vec3 v, const Vec3 a, T sine, T cosine) {
        const T k = 1 - cosine;
        return Vec3(
            v.x * (a.x * a.x + (1 - a.x * a.x) * cosine) + v.y * (a.x * a.y * k - a.z * sine) + v.z * (a.x * a.z * k + a.y * sine),
            v.x * (a.x * a.y * k + a.z * sine) + v.y * (a.y * a.y + (1 - a.y * a.y) * cosine) + v.z * (a.y * a.z * k - a.x * sine),
            v.x * (a.x * a.z * k - a.y * sine) + v.y * (a.y * a.z * k + a.x * sine) + v.z * (a.z * a.z + (1 - a.z * a.z) * cosine));
    }

    friend std::ostream &operator<<(std::ostream &os, const Vec3 &v) {
        return os << "(" << v.x << ", " << v.y << ", " << v.z << ")";
    }
};

typedef Vec3<double> Vec3d;
typedef Vec3<float> Vec3f;

}  // namespace gpuart
#endif
