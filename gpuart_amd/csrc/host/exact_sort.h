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
//   * the partitioning phase touches disjoint ranges once a range is split, so both parts can be partitioned further at the
//     same time: same comparisons, same swaps, in another order in time only. Ranges above FORK_ABOVE elements are TASKS of a
//     pool that the threads of one sort share (round 6; until then a thread handed its right part to a new thread and walked
//     the whole of a right part it could not hand over before it split its left one — 871 200 keys took 100 ms on one thread
//     and 100 ms on eight: the critical path was the whole sort. With the pool it is n + n/2 + n/4 ... comparisons).
//   * the final insertion sort runs, exactly as libstdc++ runs it (guarded for the first 16 elements, unguarded after), once
//     over the whole range on one thread IF A KEY IS NaN: a NaN compares less than nothing, partitions no longer separate
//     smaller from larger elements, and the insertion pass carries elements across partition boundaries. Primitives with NaN
//     coordinates are legal input (the reference's comparisons decide what they do), so that behaviour is reproduced too.
//     Without a NaN the ordering is valid, everything left of a partition boundary is <= everything right of it, the pass
//     never moves an element across one — its unguarded scan stops at the boundary at the latest — and it falls apart into
//     independent passes over the ranges between boundaries: each task runs its own, bounded at its first element, the
//     moment its partitioning is done (same comparisons that decide, same moves).
//
// Comparisons and moves are exactly introsort's, so ties, NaNs and adversarial inputs come out as std::sort leaves them
// (tests/test_host_parity.py::test_exact_sort_equals_std_sort holds the two against each other).
#ifndef GPUART_EXACT_SORT_H
#define GPUART_EXACT_SORT_H

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <future>
#include <mutex>
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
    /// whoever else forks work; 0: everything on this thread). Returns whether the keys are totally ordered (no NaN among
    /// them): the result is then non-decreasing from front to back.
    static bool Sort(SortKey *first, SortKey *last, std::atomic<int> &spare) {
        const ptrdiff_t n = last - first;
        if (n <= 1) return n < 1 || first->key == first->key;
        int lg = 0;
        for (size_t m = (size_t)n; m > 1; m >>= 1) lg++;
        bool has_nan = false;
        for (const SortKey *k = first; k < last; ++k) has_nan |= k->key != k->key;
        Pool pool;
        pool.array_first = first;
        pool.own_insertion = !has_nan;
        // helpers: as many as are to be had, one per FORK_ABOVE x 2 elements at most
        int helpers = 0;
        const int want = (int)std::min<ptrdiff_t>(MAX_HELPERS, n / (2 * FORK_ABOVE));
        while (helpers < want) {
            if (spare.fetch_sub(1) > 0) helpers++;
            else { spare.fetch_add(1); break; }
        }
        std::vector<std::future<void>> threads;
        pool.open = 1;
        for (int k = 0; k < helpers; k++) {
            try {
                threads.push_back(std::async(std::launch::async, [&pool] { pool.Work(); }));
            } catch (const std::system_error &) {
                spare.fetch_add(helpers - k);
                helpers = k;
                break;
            }
        }
        pool.Run(Task{first, last, 2 * lg});
        pool.Work();
        for (auto &t : threads) t.get();
        spare.fetch_add(helpers);
        if (!pool.own_insertion) {
            if (n > SMALL) {
                GuardedInsertion(first, first + SMALL);
                for (SortKey *i = first + SMALL; i < last; ++i) LinearInsert(i, first);
            } else
                GuardedInsertion(first, last);
        }
        return !has_nan;
    }
    static bool Sort(SortKey *first, SortKey *last, unsigned threads) {
        std::atomic<int> spare((int)threads - 1);
        return Sort(first, last, spare);
    }

private:
    static constexpr ptrdiff_t SMALL = 16;          ///< libstdc++'s _S_threshold
    static constexpr ptrdiff_t FORK_ABOVE = 16384;  ///< a range is a task of its own above this many elements
    static constexpr int MAX_HELPERS = 15;

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

    /// introsort's loop over [first, last) on this thread: partitions until every piece holds at most SMALL elements.
    static void SerialRange(SortKey *first, SortKey *last, int depth) {
        while (last - first > SMALL) {
            if (depth == 0) {
                std::partial_sort(first, last, last);  // libstdc++'s own heap sort of the range
                return;
            }
            --depth;
            SortKey *mid = first + (last - first) / 2;
            MedianToFront(first, first + 1, mid, last - 1);
            SortKey *cut = Partition(first + 1, last, first);
            SerialRange(cut, last, depth);
            last = cut;
        }
    }

    struct Task {
        SortKey *first, *last;
        int depth;
    };

    /// The ranges of one sort that are still to be partitioned, shared by its threads.
    struct Pool {
        std::mutex m;
        std::condition_variable cv;
        std::vector<Task> stack;
        size_t open = 0;  ///< tasks pushed or running
        SortKey *array_first = nullptr;
        bool own_insertion = false;  ///< no NaN among the keys: every range runs its own part of the final insertion pass

        /// A range that nobody splits further on another thread: introsort's loop, then — in a valid ordering — the part of the
        /// final insertion pass that falls into it. The pass as libstdc++ runs it treats the array's first SMALL positions with the
        /// guarded form and scans unguarded after; in a range that does not start the array `*i < *array_first` is false for every
        /// element and the unguarded scan stops at the range's first element at the latest (everything left of it is <=), so the
        /// bounded scan performs the same moves without reading the neighbour's elements while that one sorts them.
        void Leaf(SortKey *first, SortKey *last, int depth) {
            SerialRange(first, last, depth);
            if (!own_insertion) return;
            if (first == array_first) {
                const ptrdiff_t n = last - first;
                GuardedInsertion(first, first + std::min(n, SMALL));
                for (SortKey *i = first + std::min(n, SMALL); i < last; ++i) LinearInsert(i, first);
            } else {
                for (SortKey *i = first + 1; i < last; ++i) LinearInsert(i, first);
            }
        }

        /// Splits a task's range while it is large: the right part becomes a task of its own (or a leaf), the loop goes on left.
        void Run(Task t) {
            SortKey *first = t.first, *last = t.last;
            int depth = t.depth;
            bool heap_sorted = false;
            while (last - first > FORK_ABOVE) {
                if (depth == 0) {
                    std::partial_sort(first, last, last);
                    heap_sorted = true;  // (sorted: the insertion pass moves nothing in it)
                    break;
                }
                --depth;
                SortKey *mid = first + (last - first) / 2;
                MedianToFront(first, first + 1, mid, last - 1);
                SortKey *cut = Partition(first + 1, last, first);
                if (last - cut > FORK_ABOVE) {
                    std::lock_guard<std::mutex> g(m);
                    stack.push_back(Task{cut, last, depth});
                    open++;
                    cv.notify_one();
                } else
                    Leaf(cut, last, depth);
                last = cut;
            }
            if (!heap_sorted) Leaf(first, last, depth);
            std::lock_guard<std::mutex> g(m);
            if (--open == 0) cv.notify_all();
        }

        /// Takes tasks until none is left and none is running.
        void Work() {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [&] { return !stack.empty() || open == 0; });
                if (stack.empty()) return;
                const Task t = stack.back();
                stack.pop_back();
                lk.unlock();
                Run(t);
                lk.lock();
            }
        }
    };
};

}  // namespace gpuart
#endif
