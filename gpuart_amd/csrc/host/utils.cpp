// utils.cpp — text loaders (reference src/utils.cpp:47-203 defines the accepted dialects).
#include "utils.h"

#include <cstdarg>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

namespace gpuart {
namespace Utils {

std::ostream &operator<<(std::ostream &os, const TimeElapsed &t) {
    const std::chrono::duration<double> d = std::chrono::high_resolution_clock::now() - t.start;
    return os << d.count() << " s";
}

namespace {
/// Parses the count at the end of an "element <kind> <count>" header line.
bool header_count(const std::string &line, const char *prefix, size_t &count) {
    if (line.compare(0, std::char_traits<char>::length(prefix), prefix) != 0) return false;
    std::istringstream ss(line.substr(std::char_traits<char>::length(prefix)));
    ss >> count;
    return !ss.fail();
}
}  // namespace

bool LoadMeshFromPLY(std::vector<Primitive *> &primitives, const char *fileName, float magnification,
                     const Vec3f &translation) {
    std::ifstream fs(fileName);
    std::string line;
    if (fs.fail() || !std::getline(fs, line) || line != "ply") return false;
    const auto tstart = std::chrono::high_resolution_clock::now();
    std::cout << "Loading mesh from \"" << fileName << "\"... " << std::flush;

    size_t numVertices = 0, numFaces = 0;
    while (!fs.eof() && line != "end_header") {
        std::getline(fs, line);
        if (line.compare(0, 14, "element vertex") == 0) {
            if (!header_count(line, "element vertex", numVertices)) return false;
        } else if (line.compare(0, 12, "element face") == 0) {
            if (!header_count(line, "element face", numFaces)) return false;
        }
    }

    std::vector<Vec3f> vertices;
    vertices.reserve(numVertices);
    for (size_t i = 0; i < numVertices; i++) {
        std::getline(fs, line);
        if (line.empty()) continue;
        std::istringstream ss(line);
        float x, y, z;
        ss >> x >> y >> z;
        if (ss.fail()) return false;
        vertices.push_back(translation + magnification * Vec3f(x, y, z));
    }
    for (size_t i = 0; i < numFaces && !fs.eof(); i++) {
        std::getline(fs, line);
        if (line.empty()) continue;
        std::istringstream ss(line);
        int verts, v0, v1, v2;
        ss >> verts >> v0 >> v1 >> v2;
        if (verts != 3 || ss.fail()) return false;
        const size_t n = vertices.size();
        if (v0 < 0 || v1 < 0 || v2 < 0 || (size_t)v0 >= n || (size_t)v1 >= n || (size_t)v2 >= n) return false;
        primitives.push_back(new Triangle(vertices[v0], vertices[v1], vertices[v2]));
    }
    std::cout << " done (" << TimeElapsed(tstart) << "), faces: " << numFaces << ", vertices: " << numVertices << "."
              << std::endl;
    return true;
}

bool LoadPrimitives(std::vector<Primitive *> &primitives, const char *fileName, float magnification,
                    const Vec3f &translation) {
    std::cout << "Loading primitives from \"" << fileName << "\"..." << std::flush;
    const auto tstart = std::chrono::high_resolution_clock::now();
    std::ifstream fs(fileName);
    if (fs.fail()) return false;
    std::string line, token;
    while (std::getline(fs, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        ss >> token;
        if (token == "sphere") {
            float x, y, z, r;
            ss >> x >> y >> z;
            if (ss.fail()) return false;
            ss >> r;
            if (ss.fail()) r = 4.0f;  // the reference's default radius
            primitives.push_back(new Sphere(translation + magnification * Vec3f(x, y, z), magnification * r));
        } else if (token == "cone") {
            Vec3f c1, c2;
            float r1, r2;
            ss >> c1.x >> c1.y >> c1.z >> c2.x >> c2.y >> c2.z >> r1 >> r2;
            if (ss.fail()) return false;
            primitives.push_back(new Cone(translation + magnification * c1, translation + magnification * c2,
                                          magnification * r1, magnification * r2));
        }
    }
    std::cout << " done (" << TimeElapsed(tstart) << ")." << std::endl;
    return true;
}

std::unique_ptr<char[]> FormatStr(const char *format, ...) {
    va_list args;
    va_start(args, format);
    const int len = vsnprintf(nullptr, 0, format, args);
    va_end(args);
    std::unique_ptr<char[]> out(new char[(len < 0 ? 0 : len) + 1]);
    va_start(args, format);
    vsnprintf(out.get(), (len < 0 ? 0 : len) + 1, format, args);
    va_end(args);
    return out;
}

}  // namespace Utils
}  // namespace gpuart
