// bvh.cpp — BVH construction and compilation.
// The split rule is the reference's (src/bvh.cpp:35-152) and must stay bit-compatible with it,
// because the traversal order — and with it which of two equally distant hits wins — depends on
// the tree: longest axis of the node box (x preferred, then y), primitives sorted by box centre
// with std::sort (float sum, compared as double), split at the first centre beyond the box
// midpoint, never leaving a side empty when there are more than two primitives.
#include "bvh.h"
#include "exact_sort.h"

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <future>
#include <cstdlib>
#include <cstring>
#include <ostream>
#include <system_error>
#include <thread>
#ifdef __linux__
#include <sched.h>
#endif

namespace gpuart {

namespace {
inline float as_float(uint32_t u) { float f; std::memcpy(&f, &u, sizeof f); return f; }
inline uint32_t as_uint(float f) { uint32_t u; std::memcpy(&u, &f, sizeof u); return u; }

/// Threads a build may use: the cores this process may run on (not the machine's: a GPU box grants a share), at most 32.
int build_threads() {
    if (const char *e = std::getenv("GPUART_BVH_THREADS")) return std::max(1, std::atoi(e));
    unsigned n = std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
#endif
    return (int)std::min(32u, std::max(1u, n));
}

/// Runs body(k, parts) for k in [0, parts) on `parts` threads (the caller's included).
template <class F>
void parallel_parts(int parts, F body) {
    parts = std::max(1, parts);
    std::vector<std::future<void>> tasks;
    for (int k = 1; k < parts; k++) {
        try {
            tasks.push_back(std::async(std::launch::async, body, k, parts));
        } catch (const std::system_error &) {
            body(k, parts);
        }
    }
    body(0, parts);
    for (auto &t : tasks) t.get();
}
}  // namespace

BoundingVolumesHierarchy::BoundingVolumesHierarchy(std::vector<Primitive *> &primitives, unsigned maxNumLevels,
                                                   unsigned minPrimitivesPerNode) {
    NumPrimitives = primitives.size();
    // the build sorts (box, pointer) items instead of chasing the pointers in every comparison; std::sort's sequence of
    // moves depends only on the comparison results, so the order is the one sorting the pointers would give
    const bool timing = std::getenv("GPUART_HOST_TIMING") != nullptr;
    const auto tb = std::chrono::steady_clock::now();
    // threads are worth starting when the scene is large enough to pay for them
    const int threads = primitives.size() >= 16384 ? build_threads() : 1;
    ItemList items(primitives.size());
    parallel_parts((int)std::min<size_t>((size_t)threads, primitives.size() / 8192 + 1), [&](int k, int parts) {
        for (size_t i = primitives.size() * (size_t)k / parts, e = primitives.size() * (size_t)(k + 1) / parts; i < e; i++) {
            const Primitive *p = primitives[i];
            items[i] = Item{{p->GetXmin(), p->GetYmin(), p->GetZmin()}, {p->GetXmax(), p->GetYmax(), p->GetZmax()}, primitives[i]};
        }
    });
    Subtree root;
    std::atomic<int> spare(threads - 1);
    const auto t0 = std::chrono::steady_clock::now();
    SubdivideParallel(root, items, 0, items.size(), 0, maxNumLevels, minPrimitivesPerNode, spare);
    const auto t1 = std::chrono::steady_clock::now();
    // the reference leaves the caller's list sorted too
    parallel_parts((int)std::min<size_t>((size_t)threads, primitives.size() / 65536 + 1), [&](int k, int parts) {
        for (size_t i = items.size() * (size_t)k / parts, e = items.size() * (size_t)(k + 1) / parts; i < e; i++) primitives[i] = items[i].p;
    });
    Assemble(root, threads);
    const auto t2 = std::chrono::steady_clock::now();
    StoreLeaves(items, threads);
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[gpuart] BVH build: boxes %.1f ms, subdivide %.1f ms, assemble %.1f ms, leaf payloads %.1f ms (%d threads)\n",
                ms(tb, t0), ms(t0, t1), ms(t1, t2), ms(t2, std::chrono::steady_clock::now()), threads);
    }
}

void BoundingVolumesHierarchy::StoreLeaves(const ItemList &prims, int threads) {
    // payload length of every leaf: per primitive one type quad + 1..4 data quads (reference src/core.h:72-80)
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, Nodes.size() / 4096));
    std::vector<size_t> partLen((size_t)parts + 1, 0);
    parallel_parts(parts, [&](int k, int) {  // lengths relative to the part's start ...
        const size_t a = Nodes.size() * (size_t)k / parts, b = Nodes.size() * (size_t)(k + 1) / parts;
        size_t cursor = 0;
        for (size_t i = a; i < b; i++) {
            Node &n = Nodes[i];
            n.dataBegin = cursor;
            for (uint32_t j = 0; j < n.count; j++) cursor += prims[n.primFirst + j].p->GetBVHDataLength();
            n.dataEnd = cursor;
        }
        partLen[(size_t)k + 1] = cursor;
    });
    for (int k = 0; k < parts; k++) partLen[(size_t)k + 1] += partLen[(size_t)k];  // ... made absolute below
    LeafData.resize(partLen[(size_t)parts]);  // (not zeroed: every float is written by the part that owns it)
    parallel_parts(parts, [&](int k, int) {
        const size_t a = Nodes.size() * (size_t)k / parts, b = Nodes.size() * (size_t)(k + 1) / parts;
        for (size_t i = a; i < b; i++) { Nodes[i].dataBegin += partLen[(size_t)k]; Nodes[i].dataEnd += partLen[(size_t)k]; }
        Primitive::Data one;
        for (size_t i = a; i < b; i++) {
            const Node &n = Nodes[i];
            if (!n.count) continue;
            one.clear();
            for (uint32_t j = 0; j < n.count; j++) prims[n.primFirst + j].p->StoreIntoBVH(one);
            assert(one.size() == n.dataEnd - n.dataBegin);
            std::copy(one.begin(), one.end(), LeafData.begin() + n.dataBegin);
        }
    });
}

bool BoundingVolumesHierarchy::PrepareNode(Subtree &out, uint32_t self, ItemList &prims, size_t from, size_t to,
                                           unsigned level, unsigned maxNumLevels, unsigned minPrimitivesPerNode,
                                           size_t &split, std::atomic<int> &spareThreads, Scratch &scratch) {
    if (level > out.depth) out.depth = level;
    static const bool phases = [] { const char *e = std::getenv("GPUART_HOST_TIMING"); return e && std::atoi(e) >= 2; }();
    const auto tp0 = std::chrono::steady_clock::now();
    auto tp1 = tp0, tp2 = tp0, tp3 = tp0, tp4 = tp0;
    // the passes over a very large node's range (box, keys, permutation) run on several threads: with the subtrees assembled
    // once and the sort parallel, they are what is left of the top levels' critical path
    const int wide = to - from >= 262144 ? std::min(8, std::max(1, build_threads() / 4)) : 1;
    {
        Node &n = out.nodes[self];
        n.higher = 0; n.count = 0; n.primFirst = 0; n.dataBegin = n.dataEnd = 0;
        struct Box { float lo[3], hi[3]; };
        Box part[8];  // wide <= 8
        parallel_parts(wide, [&](int w, int) {
            Box b;
            for (int k = 0; k < 3; k++) { b.lo[k] = 99.0e+29f; b.hi[k] = -99.0e+29f; }
            const size_t a = from + (to - from) * (size_t)w / wide, e = from + (to - from) * (size_t)(w + 1) / wide;
            for (size_t i = a; i < e; i++)
                for (int k = 0; k < 3; k++) {
                    const float lo = prims[i].lo[k], hi = prims[i].hi[k];
                    if (lo < b.lo[k]) b.lo[k] = lo;
                    if (hi > b.hi[k]) b.hi[k] = hi;
                }
            part[w] = b;
        });
        // (the same running minimum / maximum: `<` and `>` keep the first of equal values and skip NaN, in parts as in one pass)
        for (int k = 0; k < 3; k++) { n.lo[k] = 99.0e+29f; n.hi[k] = -99.0e+29f; }
        for (int w = 0; w < wide; w++)
            for (int k = 0; k < 3; k++) {
                if (part[w].lo[k] < n.lo[k]) n.lo[k] = part[w].lo[k];
                if (part[w].hi[k] > n.hi[k]) n.hi[k] = part[w].hi[k];
            }
    }
    if (phases) tp1 = std::chrono::steady_clock::now();
    const float xr = out.nodes[self].hi[0] - out.nodes[self].lo[0], yr = out.nodes[self].hi[1] - out.nodes[self].lo[1],
                zr = out.nodes[self].hi[2] - out.nodes[self].lo[2];

    if (to - from <= minPrimitivesPerNode || level == maxNumLevels - 1) {
        Node &n = out.nodes[self];
        n.count = (uint32_t)(to - from);
        n.primFirst = (uint32_t)from;
        return true;
    }

    int axis;
    if (xr >= yr && xr >= zr) axis = 0;
    else if (yr >= xr && yr >= zr) axis = 1;
    else axis = 2;
    const float range = axis == 0 ? xr : axis == 1 ? yr : zr;
    // The reference sorts with std::sort on the box centres, 0.5 * (min + max) with the sum taken in float and the rest in
    // double (src/bvh.cpp:96). Comparing the float sums gives the same answers, and sorting (key, position) pairs the same
    // sequence of moves as sorting the primitives themselves; exact_sort.h performs that sort, large ones in parallel.
    bool ordered = false;  // no NaN among the keys: the sorted centres do not decrease
    {
        const size_t n = to - from;
        if (scratch.keys.size() < n) { scratch.keys.resize(n); scratch.items.resize(n); }
        SortKey *keys = scratch.keys.data();
        parallel_parts(wide, [&](int w, int) {
            for (size_t i = n * (size_t)w / wide, e = n * (size_t)(w + 1) / wide; i < e; i++)
                keys[i] = SortKey{prims[from + i].lo[axis] + prims[from + i].hi[axis], (uint32_t)i};
        });
        if (phases) tp2 = std::chrono::steady_clock::now();
        ordered = ExactSort::Sort(keys, keys + n, spareThreads);
        if (phases) tp3 = std::chrono::steady_clock::now();
        Item *sorted = scratch.items.data();
        parallel_parts(wide, [&](int w, int) {
            for (size_t i = n * (size_t)w / wide, e = n * (size_t)(w + 1) / wide; i < e; i++) sorted[i] = prims[from + keys[i].index];
        });
        parallel_parts(wide, [&](int w, int) {
            const size_t a = n * (size_t)w / wide, e = n * (size_t)(w + 1) / wide;
            std::copy(sorted + a, sorted + e, prims.begin() + from + a);
        });
    }

    if (phases) tp4 = std::chrono::steady_clock::now();
    const double middle = out.nodes[self].lo[axis] + 0.5 * range;
    // the reference scans from the front for the first centre beyond the middle (src/bvh.cpp:104-111); over centres that do not
    // decrease that is a binary search (the root of an 871 200-triangle mesh: 435 000 items of 32 bytes walked by one thread).
    // NaN centres compare false wherever they stand: then the scan itself.
    split = from;
    auto beyond = [&](size_t i) { return !(0.5 * (prims[i].lo[axis] + prims[i].hi[axis]) <= middle); };
    if (ordered) {
        size_t a = from, b = to;  // first i in [from, to] with beyond(i), to if none
        while (a < b) {
            const size_t m = a + (b - a) / 2;
            if (beyond(m)) b = m;
            else a = m + 1;
        }
        split = a;
    } else
        while (split < to && !beyond(split)) split++;
    if (to - from > 2) {  // a dominating box must not capture everything on one side
        if (split == from) split++;
        else if (split == to) split--;
    }
    if (phases && to - from >= 65536) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[gpuart] node level %u, %zu items (wide %d): box %.2f, keys %.2f, sort %.2f, permute %.2f, split %.2f ms\n", level, to - from, wide,
                ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3), ms(tp3, tp4), ms(tp4, std::chrono::steady_clock::now()));
    }
    return false;
}

void BoundingVolumesHierarchy::Subdivide(Subtree &out, ItemList &prims, size_t from, size_t to, unsigned level,
                                         unsigned maxNumLevels, unsigned minPrimitivesPerNode, uint32_t parent,
                                         bool isLower, Scratch &scratch) {
    const uint32_t self = (uint32_t)out.nodes.size();
    out.nodes.emplace_back();
    out.nodes[self].parent = parent;
    out.nodes[self].isLower = isLower;
    size_t split;
    std::atomic<int> none(0);
    if (PrepareNode(out, self, prims, from, to, level, maxNumLevels, minPrimitivesPerNode, split, none, scratch)) return;
    Subdivide(out, prims, from, split, level + 1, maxNumLevels, minPrimitivesPerNode, self, true, scratch);
    out.nodes[self].higher = (uint32_t)out.nodes.size();
    Subdivide(out, prims, split, to, level + 1, maxNumLevels, minPrimitivesPerNode, self, false, scratch);
}

void BoundingVolumesHierarchy::SubdivideParallel(Subtree &out, ItemList &prims, size_t from, size_t to,
                                                 unsigned level, unsigned maxNumLevels, unsigned minPrimitivesPerNode,
                                                 std::atomic<int> &spareThreads) {
    Scratch scratch;  // of this task: a node's sort buffers are reused by the nodes below it
    if (to - from < 8192) {
        Subdivide(out, prims, from, to, level, maxNumLevels, minPrimitivesPerNode, 0, false, scratch);
        return;
    }
    // this node (index 0 of `out`), then the two halves as independent subtrees spliced behind it in pre-order
    out.nodes.emplace_back();
    out.nodes[0].parent = 0;
    out.nodes[0].isLower = false;
    size_t split;
    if (PrepareNode(out, 0, prims, from, to, level, maxNumLevels, minPrimitivesPerNode, split, spareThreads, scratch)) return;
    { Scratch().swap(scratch); }  // the halves bring their own
    out.lo.reset(new Subtree());
    out.hi.reset(new Subtree());
    Subtree &lo = *out.lo, &hi = *out.hi;
    auto buildLo = [&] { SubdivideParallel(lo, prims, from, split, level + 1, maxNumLevels, minPrimitivesPerNode, spareThreads); };
    std::future<void> task;
    // a lopsided split (one dominating primitive) is not worth a thread
    if (std::min(split - from, to - split) >= 4096) {
        if (spareThreads.fetch_sub(1) > 0) {
            try {
                task = std::async(std::launch::async, [&] { buildLo(); spareThreads.fetch_add(1); });
            } catch (const std::system_error &) {  // no thread to be had: build this half here
                spareThreads.fetch_add(1);
            }
        } else
            spareThreads.fetch_add(1);
    }
    SubdivideParallel(hi, prims, split, to, level + 1, maxNumLevels, minPrimitivesPerNode, spareThreads);
    if (task.valid()) {
        spareThreads.fetch_add(1);  // while this thread waits, another may be started in its place
        task.get();
        spareThreads.fetch_sub(1);
    } else
        buildLo();
}

void BoundingVolumesHierarchy::Assemble(Subtree &root, int threads) {
    // pre-order walk over the pieces: where each one's first node goes
    std::vector<Subtree *> pieces;
    struct Link { Subtree *piece; size_t parentNode; bool isLower; };
    std::vector<Link> links;
    size_t total = 0;
    unsigned depth = 0;
    {
        std::vector<Link> stack{{&root, 0, false}};
        while (!stack.empty()) {
            const Link l = stack.back();
            stack.pop_back();
            l.piece->base = total;
            total += l.piece->nodes.size();
            if (l.piece->depth > depth) depth = l.piece->depth;
            links.push_back(l);
            if (l.piece->lo) {  // the upper half is laid out after the whole lower one
                stack.push_back(Link{l.piece->hi.get(), l.piece->base, false});
                stack.push_back(Link{l.piece->lo.get(), l.piece->base, true});
            }
        }
    }
    Nodes.resize(total);
    Depth = depth;
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, links.size()));
    parallel_parts(parts, [&](int k, int) {
        for (size_t li = (size_t)k; li < links.size(); li += (size_t)parts) {
            const Link &l = links[li];
            const Subtree &p = *l.piece;
            const uint32_t base = (uint32_t)p.base;
            Node *n = Nodes.data() + p.base;
            std::copy(p.nodes.begin(), p.nodes.end(), n);
            n[0].parent = (uint32_t)l.parentNode;
            n[0].isLower = l.isLower;
            if (p.lo) n[0].higher = (uint32_t)p.hi->base;  // a forked node: its upper half is a piece of its own
            else if (n[0].higher) n[0].higher += base;
            for (size_t i = 1; i < p.nodes.size(); i++) {
                n[i].parent += base;
                if (n[i].higher) n[i].higher += base;
            }
        }
    });
}

size_t BoundingVolumesHierarchy::CompiledFloats() const {
    // quad address of every node: 3 quads + its leaf payload, in pre-order (a running sum: parts sum their own nodes, the parts'
    // totals are added up, every part shifts its addresses)
    if (QuadAddr.size() == Nodes.size() + 1) return (size_t)QuadAddr.back() * RGBA_ELEMS;
    QuadAddr.resize(Nodes.size() + 1);
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)(Nodes.size() >= 65536 ? build_threads() : 1), Nodes.size() / 16384));
    std::vector<size_t> partLen((size_t)parts + 1, 0);
    parallel_parts(parts, [&](int k, int) {
        const size_t a = Nodes.size() * (size_t)k / parts, b = Nodes.size() * (size_t)(k + 1) / parts;
        size_t cursor = 0;
        for (size_t i = a; i < b; i++) {
            QuadAddr[i] = (uint32_t)cursor;
            cursor += 3 + (Nodes[i].dataEnd - Nodes[i].dataBegin) / RGBA_ELEMS;
        }
        partLen[(size_t)k + 1] = cursor;
    });
    for (int k = 0; k < parts; k++) partLen[(size_t)k + 1] += partLen[(size_t)k];
    assert(partLen[(size_t)parts] * RGBA_ELEMS <= (size_t)1 << 31);
    parallel_parts(parts, [&](int k, int) {
        if (!k) return;
        const size_t a = Nodes.size() * (size_t)k / parts, b = Nodes.size() * (size_t)(k + 1) / parts;
        for (size_t i = a; i < b; i++) QuadAddr[i] += (uint32_t)partLen[(size_t)k];
    });
    QuadAddr[Nodes.size()] = (uint32_t)partLen[(size_t)parts];
    return partLen[(size_t)parts] * RGBA_ELEMS;
}

void BoundingVolumesHierarchy::Compile(Primitive::Data &out) const {
    if (Nodes.empty()) return;
    const size_t base = out.size() / RGBA_ELEMS;
    const size_t n = CompiledFloats();
    assert((base * RGBA_ELEMS + n) <= (size_t)1 << 31);
    out.resize(base * RGBA_ELEMS + n);
    CompileTo(out.data() + base * RGBA_ELEMS, base);
}

void BoundingVolumesHierarchy::CompileTo(float *dst, size_t baseQuad) const {
    if (Nodes.empty()) return;
    CompiledFloats();
    const uint32_t base = (uint32_t)baseQuad;
    const uint32_t *addr = QuadAddr.data();
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)(Nodes.size() >= 65536 ? build_threads() : 1), Nodes.size() / 4096));
    parallel_parts(parts, [&](int k, int) {
        const size_t a = Nodes.size() * (size_t)k / parts, b = Nodes.size() * (size_t)(k + 1) / parts;
        for (size_t i = a; i < b; i++) {
            const Node &n = Nodes[i];
            float *q = dst + (size_t)addr[i] * RGBA_ELEMS;
            uint32_t flags = (n.isLower ? IS_LOWER : 0) | (i == 0 ? IS_ROOT : 0);
            const float parentBits = as_float(i == 0 ? 0u : base + addr[n.parent]);
            q[0] = n.lo[0]; q[1] = n.lo[1]; q[2] = n.lo[2]; q[3] = RGBA_PAD;
            q[4] = n.hi[0]; q[5] = n.hi[1]; q[6] = n.hi[2]; q[7] = RGBA_PAD;
            if (n.count == 0 && n.higher != 0) {
                q[8] = as_float(flags); q[9] = as_float(base + addr[i + 1]); q[10] = as_float(base + addr[n.higher]); q[11] = parentBits;
            } else {
                flags |= LEAF | (n.count & ~FLAGS_MASK);
                q[8] = as_float(flags); q[9] = RGBA_PAD; q[10] = RGBA_PAD; q[11] = parentBits;
                std::copy(LeafData.begin() + n.dataBegin, LeafData.begin() + n.dataEnd, q + 12);
            }
        }
    });
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
