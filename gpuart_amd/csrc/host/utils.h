// utils.h — loaders and small helpers of the Renderer/Scene API (reference src/utils.h:39-61).
#ifndef GPUART_UTILS_H
#define GPUART_UTILS_H

#include <chrono>
#include <iosfwd>
#include <memory>
#include <vector>

#include "core.h"
#include "math_types.h"

namespace gpuart {
namespace Utils {

/// Streams the time elapsed since `start` as "<seconds> s".
struct TimeElapsed {
    std::chrono::high_resolution_clock::time_point start;
    explicit TimeElapsed(std::chrono::high_resolution_clock::time_point s) : start(s) {}
};
std::ostream &operator<<(std::ostream &os, const TimeElapsed &t);

/// Loads an ASCII PLY triangle mesh ("ply" / "element vertex N" / "element face M" / "end_header",
/// then "x y z" and "3 a b c" lines) and appends Triangles; vertices are magnified, then translated.
bool LoadMeshFromPLY(std::vector<Primitive *> &primitives, const char *fileName, float magnification = 1.0f,
                     const Vec3f &translation = Vec3f(0, 0, 0));

/// Loads "sphere x y z [r]" / "cone x1 y1 z1 x2 y2 z2 r1 r2" lines ('#' starts a comment).
bool LoadPrimitives(std::vector<Primitive *> &primitives, const char *fileName, float magnification = 1.0f,
                    const Vec3f &translation = Vec3f(0, 0, 0));

/// printf-style formatting into a freshly allocated C string.
std::unique_ptr<char[]> FormatStr(const char *format, ...);

}  // namespace Utils
}  // namespace gpuart
#endif
