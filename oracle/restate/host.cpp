// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product.
// Host-side restatement; see host.h. Build with -ffp-contract=off (the reference is built for
// baseline x86-64, where g++ cannot contract to FMA either).
#include "host.h"

#include <math.h>

#include <algorithm>
#include <cstring>

namespace orc {

namespace {
inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

struct F3 { float x, y, z; };
inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 add(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 mul(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline F3 crs(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// math_types.h:64-68: length = sqrt(x*x + y*y + z*z); v/a multiplies by (1/a) (math_types.h:106-113)
inline float len(F3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline F3 divs(F3 a, float s) { float inv = 1 / s; return mul(a, inv); }
inline F3 unit(F3 a) { return divs(a, len(a)); }
}  // namespace

// src/core.cpp:36-65 (sphere), :80-115 (disc), core.h:119-141 + core.cpp:159-168 (triangle),
// core.cpp:191-245 (cone).
PrimRec MakePrim(const PrimDesc &d) {
    PrimRec r{};
    r.type = d.type;
    const float *f = d.f;
    switch (d.type) {
    case 0:  // sphere {c, r}
    case 1:  // disc: box is that of a sphere of the same radius
    {
        float rad = (d.type == 0) ? f[3] : f[6];
        for (int k = 0; k < 3; k++) { r.bbmin[k] = f[k] - rad; r.bbmax[k] = f[k] + rad; }
        if (d.type == 0) {
            r.ndata = 4;
            r.data[0] = f[0]; r.data[1] = f[1]; r.data[2] = f[2]; r.data[3] = rad;
        } else {
            r.ndata = 8;
            r.data[0] = f[0]; r.data[1] = f[1]; r.data[2] = f[2]; r.data[3] = rad;
            r.data[4] = f[3]; r.data[5] = f[4]; r.data[6] = f[5]; r.data[7] = 0.0f;
        }
        break;
    }
    case 2: {
        for (int k = 0; k < 3; k++) { r.bbmin[k] = 9e+19f; r.bbmax[k] = -9e+19f; }
        for (int v = 0; v < 3; v++)
            for (int k = 0; k < 3; k++) {
                float c = f[3 * v + k];
                if (c < r.bbmin[k]) r.bbmin[k] = c;
                if (c > r.bbmax[k]) r.bbmax[k] = c;
                r.data[4 * v + k] = c;
            }
        r.data[3] = r.data[7] = r.data[11] = 0.0f;
        r.ndata = 12;
        break;
    }
    case 3: {
        const float *c1 = f, *c2 = f + 3;
        float r1 = f[6], r2 = f[7];
        double d1[3] = {c1[0], c1[1], c1[2]}, d2[3] = {c2[0], c2[1], c2[2]};
        double dx = d2[0] - d1[0], dy = d2[1] - d1[1], dz = d2[2] - d1[2];
        float axisLen = (float)sqrt(dx * dx + dy * dy + dz * dz);
        double inv = 1 / (double)axisLen;
        double vd[3] = {dx * inv, dy * inv, dz * inv};
        float widthCoeff = (r2 - r1) / axisLen;
        float cosB;
        if (fabs(r1 - r2) < 1.0e-7)
            cosB = 0.0f;
        else if (r1 > r2) {
            float h = r1 * axisLen / (r1 - r2);
            cosB = (float)(r1 / sqrt((double)h * h + (double)r1 * r1));
        } else {
            float h = r2 * axisLen / (r2 - r1);
            cosB = (float)(-r2 / sqrt((double)h * h + (double)r2 * r2));
        }
        float dotAxC1 = (float)(vd[0] * d1[0] + vd[1] * d1[1] + vd[2] * d1[2]);
        for (int k = 0; k < 3; k++) {
            r.bbmin[k] = std::min(c1[k] - r1, c2[k] - r2);
            r.bbmax[k] = std::max(c1[k] + r1, c2[k] + r2);
        }
        float *o = r.data;
        o[0] = c1[0]; o[1] = c1[1]; o[2] = c1[2]; o[3] = r1;
        o[4] = c2[0]; o[5] = c2[1]; o[6] = c2[2]; o[7] = r2;
        o[8] = (float)vd[0]; o[9] = (float)vd[1]; o[10] = (float)vd[2]; o[11] = axisLen;
        o[12] = widthCoeff; o[13] = cosB; o[14] = dotAxC1; o[15] = 0.0f;
        r.ndata = 16;
        break;
    }
    }
    return r;
}

namespace {
struct Builder {
    std::vector<PrimRec> &prims;
    std::vector<const PrimRec *> order;
    std::vector<float> out;
    unsigned maxLevels, minPrims;
    int maxDepth = 0;

    void push4(float a, float b, float c, float d) { out.push_back(a); out.push_back(b); out.push_back(c); out.push_back(d); }

    // src/bvh.cpp:35-152 (Subdivide) fused with :161-222 (CompileFrom): the compile is a
    // pre-order walk, so emitting while subdividing yields the same array.
    void node(size_t from, size_t to, unsigned level, uint32_t parentAddr, bool isLower, bool isRoot) {
        if ((int)level > maxDepth) maxDepth = (int)level;
        float mn[3] = {99.0e+29f, 99.0e+29f, 99.0e+29f}, mx[3] = {-99.0e+29f, -99.0e+29f, -99.0e+29f};
        for (size_t i = from; i < to; i++)
            for (int k = 0; k < 3; k++) {
                if (order[i]->bbmin[k] < mn[k]) mn[k] = order[i]->bbmin[k];
                if (order[i]->bbmax[k] > mx[k]) mx[k] = order[i]->bbmax[k];
            }
        float range[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        uint32_t nodeAddr = (uint32_t)(out.size() / 4);
        push4(mn[0], mn[1], mn[2], 0.0f);
        push4(mx[0], mx[1], mx[2], 0.0f);
        uint32_t flags = (isLower ? (1u << 30) : 0) | (isRoot ? (1u << 29) : 0);
        if (to - from <= minPrims || level == maxLevels - 1) {
            flags |= (1u << 31) | ((uint32_t)(to - from) & ~(7u << 29));
            push4(bits2f(flags), 0.0f, 0.0f, bits2f(parentAddr));
            for (size_t i = from; i < to; i++) {
                push4(bits2f((uint32_t)order[i]->type), 0.0f, 0.0f, 0.0f);
                out.insert(out.end(), order[i]->data, order[i]->data + order[i]->ndata);
            }
            return;
        }
        int axis;
        if (range[0] >= range[1] && range[0] >= range[2]) axis = 0;
        else if (range[1] >= range[0] && range[1] >= range[2]) axis = 1;
        else axis = 2;
        std::sort(order.begin() + from, order.begin() + to, [axis](const PrimRec *a, const PrimRec *b) {
            return (a->bbmin[axis] + a->bbmax[axis]) * 0.5 < (b->bbmin[axis] + b->bbmax[axis]) * 0.5;
        });
        size_t split = from;
        while (split < to && 0.5 * (order[split]->bbmin[axis] + order[split]->bbmax[axis]) <= mn[axis] + 0.5 * range[axis])
            split++;
        if (to - from > 2) {
            if (split == from) split++;
            else if (split == to) split--;
        }
        size_t infoAt = out.size();
        push4(bits2f(flags), 0.0f, 0.0f, bits2f(parentAddr));
        out[infoAt + 1] = bits2f((uint32_t)(out.size() / 4));
        node(from, split, level + 1, nodeAddr, true, false);
        out[infoAt + 2] = bits2f((uint32_t)(out.size() / 4));
        node(split, to, level + 1, nodeAddr, false, false);
    }
};
}  // namespace

std::vector<float> BuildAndCompileBVH(std::vector<PrimRec> prims, unsigned maxNumLevels, unsigned minPrimsPerNode,
                                      int *maxDepth) {
    Builder b{prims, {}, {}, maxNumLevels, minPrimsPerNode};
    b.order.reserve(prims.size());
    for (auto &p : prims) b.order.push_back(&p);
    b.node(0, prims.size(), 0, 0, false, true);
    if (maxDepth) *maxDepth = b.maxDepth;
    return std::move(b.out);
}

// src/renderer.cpp:135-166
CameraBasis CameraScreenBasis(const float pos[3], const float dir[3], const float upv[3], float fovY, float screenDist,
                              unsigned W, unsigned H) {
    const float PI = 3.1415926f;
    F3 P{pos[0], pos[1], pos[2]}, D{dir[0], dir[1], dir[2]}, U{upv[0], upv[1], upv[2]};
    float aspect = (float)W / H;
    F3 up = unit(crs(crs(D, U), D));
    F3 target = add(P, mul(unit(D), screenDist));
    F3 a = mul(mul(mul(crs(unit(D), up), screenDist), aspect), tan(fovY / 2 * PI / 180));
    F3 b = divs(mul(up, len(a)), aspect);
    F3 bl = sub(sub(target, a), b);
    F3 dh = mul(a, 2), dv = mul(b, 2);
    CameraBasis r;
    r.pos[0] = P.x; r.pos[1] = P.y; r.pos[2] = P.z;
    r.bottomLeft[0] = bl.x; r.bottomLeft[1] = bl.y; r.bottomLeft[2] = bl.z;
    r.deltaHorz[0] = dh.x; r.deltaHorz[1] = dh.y; r.deltaHorz[2] = dh.z;
    r.deltaVert[0] = dv.x; r.deltaVert[1] = dv.y; r.deltaVert[2] = dv.z;
    return r;
}

// src/renderer.cpp:573-574
float PixelSize(float fovY, float screenDist, unsigned H) {
    const float PI = 3.1415926f;
    return 2 * screenDist * tan(fovY / 2 * PI / 180) / H;
}

// src/renderer.h:175-179 with math_types.h vroty/vrotz (float cos/sin)
void SunDirection(float az, float alt, float out[3]) {
    float a = -alt;
    float c = cos(a), s = sin(a);
    F3 v{0 * s + 1 * c, 0, 0 * c - 1 * s};
    float cz = cos(az), sz = sin(az);
    out[0] = v.x * cz - v.y * sz;
    out[1] = v.x * sz + v.y * cz;
    out[2] = v.z;
}

}  // namespace orc
