// Exact restatement of libstdc++'s std::sort (introsort, GCC 4.x .. 13: bits/stl_algo.h __introsort_loop /
// __unguarded_partition_pivot / __final_insertion_sort, bits/stl_heap.h for the depth-limit fallback) so that the device can
// reproduce the ORDER the reference gets from its unstable `std::sort` of the candidate list (ref actor/zero_actor.cpp:225-227)
// even when policies tie.  The algorithm is sequential; on the device one lane runs it over LDS, and only when a tie exists
// among more than 16 candidates (without ties every correct sort gives the same order; up to 16 elements std::sort is a
// stable insertion sort).  tests/test_sort_emul.py compiles this header with g++ and checks it against the real std::sort /
// std::partial_sort on tie-heavy inputs.
#pragma once

#ifdef __HIPCC__
#define MZ_HD __host__ __device__
#else
#define MZ_HD
#endif

namespace mz {

// Up to _S_threshold = 16 elements std::sort is __insertion_sort alone (bits/stl_algo.h __final_insertion_sort), a STABLE sort: element i ends up behind
// the elements that come before it under comp and behind the equivalent ones it followed.  That position can be counted for every element independently —
// what the device's wave-parallel sorts of <= 16 Gumbel candidates do (gumbel_body.h sortSmallStable); checked against the real std::sort by
// tests/test_sort_emul.py.
constexpr int kStdSortInsertionOnly = 16;
template <class T, class Comp>
MZ_HD int stableRankOf(const T* a, int n, int i, Comp comp)
{
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        if (j != i) { rank += (j < i ? !comp(a[i], a[j]) : comp(a[j], a[i])) ? 1 : 0; }
    }
    return rank;
}

// A = random-access "array view" with T get(i) / void set(i, T) is overkill here: the candidates are a plain struct array.
template <class T, class Comp>
struct StdSortEmul {
    T* a;
    Comp comp;

    MZ_HD void swapAt(int i, int j) { T t = a[i]; a[i] = a[j]; a[j] = t; }

    // ---- bits/stl_heap.h ----
    MZ_HD void pushHeap(int first, int hole, int top, T value)
    {
        int parent = (hole - 1) / 2;
        while (hole > top && comp(a[first + parent], value)) {
            a[first + hole] = a[first + parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        a[first + hole] = value;
    }
    MZ_HD void adjustHeap(int first, int hole, int len, T value)
    {
        const int top = hole;
        int second = hole;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (comp(a[first + second], a[first + (second - 1)])) { --second; }
            a[first + hole] = a[first + second];
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            a[first + hole] = a[first + (second - 1)];
            hole = second - 1;
        }
        pushHeap(first, hole, top, value);
    }
    MZ_HD void makeHeap(int first, int last)
    {
        const int len = last - first;
        if (len < 2) { return; }
        int parent = (len - 2) / 2;
        while (true) {
            T value = a[first + parent];
            adjustHeap(first, parent, len, value);
            if (parent == 0) { return; }
            --parent;
        }
    }
    // std::partial_sort(first, last, last): __heap_select (make_heap; nothing beyond middle) + __sort_heap
    MZ_HD void heapSort(int first, int last)
    {
        makeHeap(first, last);
        while (last - first > 1) {
            --last;
            T value = a[last]; // __pop_heap(first, last, last)
            a[last] = a[first];
            adjustHeap(first, 0, last - first, value);
        }
    }

    // ---- bits/stl_algo.h ----
    MZ_HD void moveMedianToFirst(int result, int x, int y, int z)
    {
        if (comp(a[x], a[y])) {
            if (comp(a[y], a[z])) { swapAt(result, y); }
            else if (comp(a[x], a[z])) { swapAt(result, z); }
            else { swapAt(result, x); }
        } else if (comp(a[x], a[z])) { swapAt(result, x); }
        else if (comp(a[y], a[z])) { swapAt(result, z); }
        else { swapAt(result, y); }
    }
    MZ_HD int unguardedPartition(int first, int last, int pivot)
    {
        while (true) {
            while (comp(a[first], a[pivot])) { ++first; }
            --last;
            while (comp(a[pivot], a[last])) { --last; }
            if (!(first < last)) { return first; }
            swapAt(first, last);
            ++first;
        }
    }
    MZ_HD void unguardedLinearInsert(int last)
    {
        T val = a[last];
        int next = last - 1;
        while (comp(val, a[next])) {
            a[last] = a[next];
            last = next;
            --next;
        }
        a[last] = val;
    }
    MZ_HD void insertionSort(int first, int last)
    {
        if (first == last) { return; }
        for (int i = first + 1; i != last; ++i) {
            if (comp(a[i], a[first])) {
                T val = a[i];
                for (int j = i; j > first; --j) { a[j] = a[j - 1]; } // move_backward(first, i, i + 1)
                a[first] = val;
            } else {
                unguardedLinearInsert(i);
            }
        }
    }
    MZ_HD static int lg(int n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; } // std::__lg

    static constexpr int kStack = 40; // one pending range per introsort level: 2 * lg(n) levels at most
    // std::sort(a, a + n, comp).  `stack` = 3 * kStack ints of scratch (LDS on the device).  Returns false if the range stack
    // overflowed (cannot happen for n < 2^20).
    MZ_HD bool sort(int n, int* stack)
    {
        constexpr int kThreshold = 16;
        if (n <= 0) { return true; }
        // __introsort_loop recurses on [cut, last) and loops on [first, cut): disjoint ranges, so an explicit stack of the
        // pending right halves (with the depth budget they were given) reproduces the same element moves
        int *sf = stack, *sl = stack + kStack, *sd = stack + 2 * kStack, sp = 0;
        int first = 0, last = n, depth = lg(n) * 2;
        while (true) {
            while (last - first > kThreshold) {
                if (depth == 0) { heapSort(first, last); break; }
                --depth;
                const int mid = first + (last - first) / 2;
                moveMedianToFirst(first, first + 1, mid, last - 1);
                const int cut = unguardedPartition(first + 1, last, first);
                if (sp == kStack) { return false; }
                sf[sp] = cut; sl[sp] = last; sd[sp] = depth; ++sp;
                last = cut;
            }
            if (sp == 0) { break; }
            --sp;
            first = sf[sp]; last = sl[sp]; depth = sd[sp];
        }
        // __final_insertion_sort
        if (n > kThreshold) {
            insertionSort(0, kThreshold);
            for (int i = kThreshold; i != n; ++i) { unguardedLinearInsert(i); }
        } else {
            insertionSort(0, n);
        }
        return true;
    }
};

} // namespace mz
