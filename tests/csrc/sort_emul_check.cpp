// Checks minizero_amd/csrc/sort_emul.h against the real libstdc++ std::sort / std::partial_sort (tests/test_sort_emul.py).
#include "../../minizero_amd/csrc/sort_emul.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

struct Cand { int action; float policy, logit; };
struct Greater { bool operator()(const Cand& l, const Cand& r) const { return l.policy > r.policy; } };

int main()
{
    std::mt19937 gen(12345);
    long cases = 0, tie_cases = 0;
    for (int iter = 0; iter < 60000; ++iter) {
        const int n = 1 + gen() % (iter % 7 == 0 ? 400 : 90);
        const int levels = 1 + gen() % (iter % 3 == 0 ? 3 : 40); // few distinct values -> many ties
        std::vector<Cand> a(n);
        for (int i = 0; i < n; ++i) { a[i] = Cand{i, static_cast<float>(gen() % levels) / levels, static_cast<float>(i)}; }
        if (iter % 11 == 0) { std::sort(a.begin(), a.end(), [](const Cand& l, const Cand& r) { return l.policy < r.policy; }); } // organ-pipe-ish / sorted inputs
        if (iter % 13 == 0) { for (int i = 0; i < n; ++i) { a[i].policy = (i < n / 2) ? float(i) : float(n - i); } }
        std::vector<Cand> ref = a, mine = a;
        std::sort(ref.begin(), ref.end(), Greater());
        mz::StdSortEmul<Cand, Greater> s{mine.data(), Greater()};
        int stack[3 * mz::StdSortEmul<Cand, Greater>::kStack];
        if (!s.sort(n, stack)) { printf("FAIL stack overflow n=%d\n", n); return 1; }
        for (int i = 0; i < n; ++i) {
            if (ref[i].action != mine[i].action) { printf("FAIL sort iter=%d n=%d at %d: %d vs %d\n", iter, n, i, ref[i].action, mine[i].action); return 1; }
        }
        ++cases;
        tie_cases += levels < n;
        // the depth-limit fallback: std::partial_sort(first, last, last) is exactly what __introsort_loop calls
        std::vector<Cand> r2 = a, m2 = a;
        std::partial_sort(r2.begin(), r2.end(), r2.end(), Greater());
        mz::StdSortEmul<Cand, Greater> h{m2.data(), Greater()};
        h.heapSort(0, n);
        for (int i = 0; i < n; ++i) {
            if (r2[i].action != m2[i].action) { printf("FAIL heap iter=%d n=%d at %d\n", iter, n, i); return 1; }
        }
    }
    printf("OK %ld cases (%ld with ties)\n", cases, tie_cases);
    return 0;
}
