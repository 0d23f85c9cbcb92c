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
    // <= 16 elements: std::sort is the stable insertion sort, whose result the rank count of stableRankOf gives (the wave-parallel Gumbel sorts);
    // the Gumbel comparators' shape: a primary key with many ties and a secondary key, as indices into key arrays
    long small = 0;
    for (int iter = 0; iter < 40000; ++iter) {
        const int n = 1 + gen() % mz::kStdSortInsertionOnly;
        const int levels = 1 + gen() % 4;
        float cnt[16], lg[16];
        for (int i = 0; i < 16; ++i) { cnt[i] = static_cast<float>(gen() % levels); lg[i] = static_cast<float>(gen() % 3); }
        std::vector<int> ids(n);
        for (int i = 0; i < n; ++i) { ids[i] = static_cast<int>(gen() % 16); } // (duplicates on purpose: equivalent elements that are not identical positions)
        auto byCount = [&](int l, int r) { return cnt[l] < cnt[r] || (cnt[l] == cnt[r] && lg[l] > lg[r]); };
        std::vector<int> ref = ids, mine(n);
        // tag every element with its original position so that the permutation (not just the keys) is compared
        std::vector<std::pair<int, int>> tagged(n), tref;
        for (int i = 0; i < n; ++i) { tagged[i] = {ids[i], i}; }
        tref = tagged;
        auto byCountT = [&](const std::pair<int, int>& l, const std::pair<int, int>& r) { return byCount(l.first, r.first); };
        std::sort(tref.begin(), tref.end(), byCountT);
        std::vector<std::pair<int, int>> tm(n);
        for (int i = 0; i < n; ++i) { tm[mz::stableRankOf(tagged.data(), n, i, byCountT)] = tagged[i]; }
        for (int i = 0; i < n; ++i) {
            if (tm[i] != tref[i]) { printf("FAIL small iter=%d n=%d at %d\n", iter, n, i); return 1; }
        }
        ++small;
    }
    printf("OK %ld cases (%ld with ties), %ld small stable cases\n", cases, tie_cases, small);
    return 0;
}
