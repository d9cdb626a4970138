// exact_sort.h — the permutation libstdc++'s std::sort would produce, computed in parallel.
//
// The reference sorts the primitives of every BVH node with std::sort (src/bvh.cpp:96). std::sort is not stable, box
// centres tie all the time in meshes (two triangles of a quad share their extreme vertices along an axis), and the order
// of tied primitives decides which of them shares a leaf — so the tree, and with it the traversal order the images
// depend on bit for bit, is a function of libstdc++'s introsort, not just of the keys. To build the same tree faster the
// sort is therefore not replaced but re-scheduled:
//
//   * introsort's shape: while a range holds more than 16 elements — when the depth budget 2*floor(log2 n) is used up,
//     heap-sort it (std::partial_sort over the whole range, the very call libstdc++ makes); else move the median of
//     {first+1, middle, last-1} to the front, partition the rest around it (Hoare, unguarded), recurse into the right part,
//     continue with the left. Afterwards one insertion sort over everything.
//   * the partitioning phase touches disjoint ranges once a range is split, so the right part can be partitioned further by
//     another thread while this one continues with the left: same comparisons, same swaps, in another order in time only.
//   * the final insertion sort then runs once over the whole range on one thread, exactly as libstdc++ runs it (guarded
//     for the first 16 elements, unguarded after). It is cheap — with a valid ordering no element moves further than its
//     leftover range of <= 16 — and it must stay whole: keys that are NaN break the ordering (a NaN compares less than
//     nothing), partitions no longer separate smaller from larger elements, and the insertion pass carries elements across
//     partition boundaries. Primitives with NaN coordinates are legal input (the reference's comparisons decide what they
//     do), so that behaviour is reproduced too.
//
// Comparisons and moves are exactly introsort's, so ties, NaNs and adversarial inputs come out as std::sort leaves them
// (tests/test_host_parity.py::test_exact_sort_equals_std_sort holds the two against each other).
#ifndef GPUART_EXACT_SORT_H
#define GPUART_EXACT_SORT_H

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <future>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

namespace gpuart {

/// What is sorted: the primitive's key along the split axis (float sum of box min and max: comparing the sums is comparing
/// the reference's double centres, 0.5 * sum) and its position in the node's range.
struct SortKey {
    float key;
    uint32_t index;
};
inline bool operator<(const SortKey &a, const SortKey &b) { return a.key < b.key; }

class ExactSort {
public:
    /// Sorts [first, last) as std::sort(first, last) would. `spare` = threads that may still be started (shared with
    /// whoever else forks work; 0: everything on this thread).
    static void Sort(SortKey *first, SortKey *last, std::atomic<int> &spare) {
        const ptrdiff_t n = last - first;
        if (n <= 1) return;
        int lg = 0;
        for (size_t m = (size_t)n; m > 1; m >>= 1) lg++;
        Range(first, last, 2 * lg, spare);
        if (n > SMALL) {
            GuardedInsertion(first, first + SMALL);
            for (SortKey *i = first + SMALL; i < last; ++i) LinearInsert(i, first);
        } else
            GuardedInsertion(first, last);
    }
    static void Sort(SortKey *first, SortKey *last, unsigned threads) {
        std::atomic<int> spare((int)threads - 1);
        Sort(first, last, spare);
    }

private:
    static constexpr ptrdiff_t SMALL = 16;          ///< libstdc++'s _S_threshold
    static constexpr ptrdiff_t FORK_ABOVE = 16384;  ///< a right part is worth a thread above this many elements

    static void MedianToFront(SortKey *result, SortKey *a, SortKey *b, SortKey *c) {
        if (*a < *b) {
            if (*b < *c) std::swap(*result, *b);
            else if (*a < *c) std::swap(*result, *c);
            else std::swap(*result, *a);
        } else if (*a < *c) std::swap(*result, *a);
        else if (*b < *c) std::swap(*result, *c);
        else std::swap(*result, *b);
    }

    /// Hoare partition of [first, last) around *pivot (which sits just before `first`); both scans rely on sentinels
    /// the median selection left behind.
    static SortKey *Partition(SortKey *first, SortKey *last, const SortKey *pivot) {
        for (;;) {
            while (*first < *pivot) ++first;
            --last;
            while (*pivot < *last) --last;
            if (!(first < last)) return first;
            std::swap(*first, *last);
            ++first;
        }
    }

    /// Moves *at left past every element it compares less than. libstdc++ has no lower bound here (a smaller-or-equal
    /// element to the left is guaranteed by a valid ordering); `bound` only stops what would be undefined behaviour there.
    static void LinearInsert(SortKey *at, SortKey *bound) {
        const SortKey v = *at;
        SortKey *hole = at;
        while (hole > bound && v < hole[-1]) { *hole = hole[-1]; --hole; }
        *hole = v;
    }

    /// libstdc++'s insertion sort of a range that has no sentinel to its left: an element smaller than the FIRST goes
    /// straight to the front (whatever lies between — with NaN keys that is not the same as scanning), others are
    /// inserted linearly.
    static void GuardedInsertion(SortKey *first, SortKey *last) {
        if (first == last) return;
        for (SortKey *i = first + 1; i < last; ++i) {
            if (*i < *first) {
                const SortKey v = *i;
                std::move_backward(first, i, i + 1);
                *first = v;
            } else
                LinearInsert(i, first);
        }
    }

    static void Range(SortKey *first, SortKey *last, int depth, std::atomic<int> &spare) {
        std::vector<std::future<void>> forked;
        while (last - first > SMALL) {
            if (depth == 0) {
                std::partial_sort(first, last, last);  // libstdc++'s own heap sort of the range
                break;
            }
            --depth;
            SortKey *mid = first + (last - first) / 2;
            MedianToFront(first, first + 1, mid, last - 1);
            SortKey *cut = Partition(first + 1, last, first);
            bool handed_over = false;
            if (last - cut > FORK_ABOVE && cut - first > FORK_ABOVE && spare.fetch_sub(1) > 0) {
                try {
                    forked.push_back(std::async(std::launch::async, [cut, last, depth, &spare] {
                        Range(cut, last, depth, spare);
                        spare.fetch_add(1);
                    }));
                    handed_over = true;
                } catch (const std::system_error &) {
                    spare.fetch_add(1);
                }
            } else if (last - cut > FORK_ABOVE && cut - first > FORK_ABOVE) {
                spare.fetch_add(1);  // undo the probe: no thread was free
            }
            if (!handed_over) Range(cut, last, depth, spare);
            last = cut;
        }
        for (auto &f : forked) f.get();
    }
};

}  // namespace gpuart
#endif
