// bvh.h — AABB bounding-volume hierarchy of the Renderer/Scene API.
// Public surface as in the reference (src/bvh.h:46-93): construct from a primitive list with
// (maxNumLevels, minPrimitivesPerNode), Compile() into the canonical RGBA32F quad array, Print().
// Internals are different: the tree is a flat pre-order node vector that already owns the
// serialised leaf payloads, so nothing dangles after the caller deletes its primitives.
#ifndef GPUART_BVH_H
#define GPUART_BVH_H

#include <atomic>
#include <cstdint>
#include <iosfwd>
#include <memory>
#include <vector>

#include "core.h"

namespace gpuart {

struct SortKey;

class BoundingVolumesHierarchy {
public:
    // Node flags of the compiled format (reference src/bvh.h:48-52)
    static const uint32_t LEAF = 1u << 31;
    static const uint32_t IS_LOWER = 1u << 30;
    static const uint32_t IS_ROOT = 1u << 29;
    static const uint32_t FLAGS_MASK = LEAF | IS_LOWER | IS_ROOT;

    BoundingVolumesHierarchy() = default;
    BoundingVolumesHierarchy(const BoundingVolumesHierarchy &) = delete;
    BoundingVolumesHierarchy &operator=(const BoundingVolumesHierarchy &) = delete;
    BoundingVolumesHierarchy(BoundingVolumesHierarchy &&) = default;
    BoundingVolumesHierarchy &operator=(BoundingVolumesHierarchy &&) = default;

    /// Builds the tree. The order of elements in `primitives` may change.
    BoundingVolumesHierarchy(std::vector<Primitive *> &primitives, unsigned maxNumLevels, unsigned minPrimitivesPerNode);

    /// Appends the compiled tree to `compiledTree`:
    ///   node      = {xmin,ymin,zmin,pad}{xmax,ymax,zmax,pad}{flags|count, lo, hi, parent}   (uint bits in floats,
    ///               addresses in quads; the lower child follows its parent; leaves have lo = hi = 0.0f)
    ///   leaf data = per primitive {type,pad,pad,pad} + payload (Primitive::StoreIntoBVH)
    void Compile(Primitive::Data &compiledTree) const;
    /// The same quads without the vector (whose resize zeroes 105 MB on one thread for an 871 200-triangle mesh before the threads
    /// that fill it get to touch it): how many floats the compiled tree takes, and the tree written to `dst` (that many floats,
    /// uninitialised memory is fine; `baseQuad` = the quad address the tree starts at, 0 for a tree on its own).
    size_t CompiledFloats() const;
    void CompileTo(float *dst, size_t baseQuad = 0) const;

    /// Prints a compiled tree, decoding it the way the device code does.
    static void Print(const Primitive::Data &compiledTree, std::ostream &s);

    size_t GetNumNodes() const { return Nodes.size(); }
    size_t GetNumPrimitives() const { return NumPrimitives; }
    unsigned GetDepth() const { return Depth; }  ///< level of the deepest node (root = 0)

private:
    struct Node {
        float lo[3], hi[3];
        uint32_t parent;        ///< index of the parent node (root: 0)
        uint32_t higher;        ///< index of the upper child; 0 for leaves (the lower child is index+1)
        bool isLower;
        uint32_t count;         ///< primitives in a leaf (0 = interior node)
        uint32_t primFirst;     ///< leaf: position of its first primitive in the sorted primitive list
        size_t dataBegin, dataEnd;  ///< leaf payload range in LeafData (floats); filled in once the tree is complete
    };
    /// A primitive during the build: its box and itself.
    struct Item {
        float lo[3], hi[3];
        Primitive *p;
    };
    /// std::allocator whose value-less construct() leaves trivial elements uninitialised: a large buffer is then touched for the
    /// first time by the threads that fill it, not zeroed by one thread beforehand.
    template <class T>
    struct UninitAlloc : std::allocator<T> {
        template <class U> struct rebind { using other = UninitAlloc<U>; };
        template <class U> void construct(U *p) { ::new ((void *)p) U; }
        template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) { ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
    };
    using ItemList = std::vector<Item, UninitAlloc<Item>>;  ///< (filled by the threads that compute the boxes, not zeroed first)
    /// Sort buffers of one build task (exact_sort.h keys, the permuted primitives), grown on demand.
    struct Scratch {  // (not zeroed when they grow: every element in use is written first, by the threads that fill it)
        std::vector<SortKey, UninitAlloc<SortKey>> keys;
        ItemList items;
        void swap(Scratch &o) { keys.swap(o.keys); items.swap(o.items); }
    };
    /// A subtree under construction: nodes in pre-order with indices relative to the subtree (leaves refer to their
    /// primitives by position in the list: a subtree only ever reorders its own range of it).
    struct Subtree {
        std::vector<Node> nodes;
        unsigned depth = 0;  ///< deepest level reached (absolute)
        /// A large node built in parallel keeps only itself in `nodes`; its halves are subtrees of their own, put together
        /// once, at the end (Assemble): nothing is copied level by level.
        std::unique_ptr<Subtree> lo, hi;
        size_t base = 0;     ///< index of nodes[0] in the finished tree (Assemble)
    };
    /// Lays the pieces of a parallel build out in pre-order as `Nodes` (indices made absolute; several threads).
    void Assemble(Subtree &root, int threads);
    /// Serialises the primitives of all leaves into LeafData (Primitive::StoreIntoBVH), several threads on node ranges.
    void StoreLeaves(const ItemList &prims, int threads);
    std::vector<Node, UninitAlloc<Node>> Nodes;  ///< pre-order (every field of every node is written by Assemble's threads)
    mutable std::vector<uint32_t, UninitAlloc<uint32_t>> QuadAddr;  ///< quad address of every node relative to the tree's first quad (CompiledFloats)
    std::vector<float, UninitAlloc<float>> LeafData;  ///< serialised primitives of all leaves, in leaf order
    size_t NumPrimitives = 0;
    unsigned Depth = 0;

    /// Sequential build of [from, to) appended to `out` (reference src/bvh.cpp:35-152).
    static void Subdivide(Subtree &out, ItemList &prims, size_t from, size_t to, unsigned level,
                          unsigned maxNumLevels, unsigned minPrimitivesPerNode, uint32_t parent, bool isLower, Scratch &scratch);
    /// The same tree, built in parallel: the two halves of large nodes by different threads, and the sort
    /// of a large node itself by several (exact_sort.h: libstdc++'s introsort, re-scheduled). `spareThreads` = threads
    /// that may still be started, shared by both. Byte-identical to the sequential build: every node's permutation is
    /// the one std::sort produces, the halves work on disjoint ranges of `prims`.
    static void SubdivideParallel(Subtree &out, ItemList &prims, size_t from, size_t to, unsigned level,
                                  unsigned maxNumLevels, unsigned minPrimitivesPerNode, std::atomic<int> &spareThreads);
    /// Fills node `self` of `out` with the box of [from, to); returns true if it became a leaf, else sorts the range
    /// along the split axis and returns the split position.
    static bool PrepareNode(Subtree &out, uint32_t self, ItemList &prims, size_t from, size_t to, unsigned level,
                            unsigned maxNumLevels, unsigned minPrimitivesPerNode, size_t &split, std::atomic<int> &spareThreads,
                            Scratch &scratch);
};

}  // namespace gpuart
#endif
