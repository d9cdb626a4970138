// bvh.cpp — BVH construction and compilation.
// The split rule is the reference's (src/bvh.cpp:35-152) and must stay bit-compatible with it,
// because the traversal order — and with it which of two equally distant hits wins — depends on
// the tree: longest axis of the node box (x preferred, then y), primitives sorted by box centre
// with std::sort (float sum, compared as double), split at the first centre beyond the box
// midpoint, never leaving a side empty when there are more than two primitives.
#include "bvh.h"

#include <algorithm>
#include <cassert>
#include <cstring>
#include <ostream>

namespace gpuart {

namespace {
inline float as_float(uint32_t u) { float f; std::memcpy(&f, &u, sizeof f); return f; }
inline uint32_t as_uint(float f) { uint32_t u; std::memcpy(&u, &f, sizeof u); return u; }

inline float box_lo(const Primitive *p, int axis) { return axis == 0 ? p->GetXmin() : axis == 1 ? p->GetYmin() : p->GetZmin(); }
inline float box_hi(const Primitive *p, int axis) { return axis == 0 ? p->GetXmax() : axis == 1 ? p->GetYmax() : p->GetZmax(); }

template <int AXIS>
struct CentreLess {
    bool operator()(const Primitive *a, const Primitive *b) const {
        return (box_lo(a, AXIS) + box_hi(a, AXIS)) * 0.5 < (box_lo(b, AXIS) + box_hi(b, AXIS)) * 0.5;
    }
};
}  // namespace

BoundingVolumesHierarchy::BoundingVolumesHierarchy(std::vector<Primitive *> &primitives, unsigned maxNumLevels,
                                                   unsigned minPrimitivesPerNode) {
    NumPrimitives = primitives.size();
    Nodes.reserve(primitives.size() + 1);
    Subdivide(primitives, 0, primitives.size(), 0, maxNumLevels, minPrimitivesPerNode, 0, false);
}

void BoundingVolumesHierarchy::Subdivide(std::vector<Primitive *> &prims, size_t from, size_t to, unsigned level,
                                         unsigned maxNumLevels, unsigned minPrimitivesPerNode, uint32_t parent,
                                         bool isLower) {
    const uint32_t self = (uint32_t)Nodes.size();
    Nodes.emplace_back();
    if (level > Depth) Depth = level;
    {
        Node &n = Nodes.back();
        n.parent = parent; n.isLower = isLower; n.higher = 0; n.count = 0; n.dataBegin = n.dataEnd = 0;
        for (int k = 0; k < 3; k++) { n.lo[k] = 99.0e+29f; n.hi[k] = -99.0e+29f; }
        for (size_t i = from; i < to; i++)
            for (int k = 0; k < 3; k++) {
                float lo = box_lo(prims[i], k), hi = box_hi(prims[i], k);
                if (lo < n.lo[k]) n.lo[k] = lo;
                if (hi > n.hi[k]) n.hi[k] = hi;
            }
    }
    const float xr = Nodes[self].hi[0] - Nodes[self].lo[0], yr = Nodes[self].hi[1] - Nodes[self].lo[1],
                zr = Nodes[self].hi[2] - Nodes[self].lo[2];

    if (to - from <= minPrimitivesPerNode || level == maxNumLevels - 1) {
        Node &n = Nodes[self];
        n.count = (uint32_t)(to - from);
        n.dataBegin = LeafData.size();
        for (size_t i = from; i < to; i++) prims[i]->StoreIntoBVH(LeafData);
        n.dataEnd = LeafData.size();
        return;
    }

    int axis;
    if (xr >= yr && xr >= zr) axis = 0;
    else if (yr >= xr && yr >= zr) axis = 1;
    else axis = 2;
    const float range = axis == 0 ? xr : axis == 1 ? yr : zr;
    auto first = prims.begin() + from, last = prims.begin() + to;
    if (axis == 0) std::sort(first, last, CentreLess<0>());
    else if (axis == 1) std::sort(first, last, CentreLess<1>());
    else std::sort(first, last, CentreLess<2>());

    const double middle = Nodes[self].lo[axis] + 0.5 * range;
    size_t split = from;
    while (split < to && 0.5 * (box_lo(prims[split], axis) + box_hi(prims[split], axis)) <= middle) split++;
    if (to - from > 2) {  // a dominating box must not capture everything on one side
        if (split == from) split++;
        else if (split == to) split--;
    }

    Subdivide(prims, from, split, level + 1, maxNumLevels, minPrimitivesPerNode, self, true);
    Nodes[self].higher = (uint32_t)Nodes.size();
    Subdivide(prims, split, to, level + 1, maxNumLevels, minPrimitivesPerNode, self, false);
}

void BoundingVolumesHierarchy::Compile(Primitive::Data &out) const {
    if (Nodes.empty()) return;
    // quad address of every node: 3 quads + its leaf payload, in pre-order
    const size_t base = out.size() / RGBA_ELEMS;
    std::vector<uint32_t> addr(Nodes.size());
    size_t cursor = base;
    for (size_t i = 0; i < Nodes.size(); i++) {
        addr[i] = (uint32_t)cursor;
        cursor += 3 + (Nodes[i].dataEnd - Nodes[i].dataBegin) / RGBA_ELEMS;
    }
    assert(cursor * RGBA_ELEMS <= (size_t)1 << 31);
    out.reserve(cursor * RGBA_ELEMS);
    for (size_t i = 0; i < Nodes.size(); i++) {
        const Node &n = Nodes[i];
        uint32_t flags = (n.isLower ? IS_LOWER : 0) | (i == 0 ? IS_ROOT : 0);
        const float parentBits = as_float(i == 0 ? 0u : addr[n.parent]);
        const float quads[8] = {n.lo[0], n.lo[1], n.lo[2], RGBA_PAD, n.hi[0], n.hi[1], n.hi[2], RGBA_PAD};
        out.insert(out.end(), quads, quads + 8);
        if (n.count == 0 && n.higher != 0) {
            const float info[4] = {as_float(flags), as_float(addr[i + 1]), as_float(addr[n.higher]), parentBits};
            out.insert(out.end(), info, info + 4);
        } else {
            flags |= LEAF | (n.count & ~FLAGS_MASK);
            const float info[4] = {as_float(flags), RGBA_PAD, RGBA_PAD, parentBits};
            out.insert(out.end(), info, info + 4);
            out.insert(out.end(), LeafData.begin() + n.dataBegin, LeafData.begin() + n.dataEnd);
        }
    }
}

void BoundingVolumesHierarchy::Print(const Primitive::Data &t, std::ostream &s) {
    auto it = t.begin();
    while (it != t.end()) {
        const size_t a = (it - t.begin()) / RGBA_ELEMS;
        s << "[" << a << "] box (" << it[0] << ", " << it[1] << ", " << it[2] << ") - (" << it[4] << ", " << it[5] << ", "
          << it[6] << ")";
        const uint32_t flags = as_uint(it[8]);
        s << ((flags & IS_ROOT) ? " ROOT" : "") << ((flags & IS_LOWER) ? " LOWER" : "") << " parent " << as_uint(it[11]);
        if (flags & LEAF) {
            const uint32_t n = flags & ~FLAGS_MASK;
            s << " LEAF x" << n << ":";
            it += 12;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t type = as_uint(*it);
                it += RGBA_ELEMS;
                switch (type) {
                case SPHERE: s << " sphere "; Sphere::PrintBVH(it, s); break;
                case DISC: s << " disc "; Disc::PrintBVH(it, s); break;
                case TRIANGLE: s << " triangle "; Triangle::PrintBVH(it, s); break;
                case CONE: s << " cone "; Cone::PrintBVH(it, s); break;
                default: s << " <unknown primitive type " << type << ">\n"; return;
                }
            }
        } else {
            s << " lo " << as_uint(it[9]) << " hi " << as_uint(it[10]);
            it += 12;
        }
        s << "\n";
    }
}

}  // namespace gpuart
