// The real libstdc++ std::sort on a candidate list (policy descending), for the device-sort GPU test.
#include <algorithm>
#include <vector>
struct Cand { int action; float policy, logit; };
extern "C" void ref_sort(const float* policy, int n, int* order)
{
    std::vector<Cand> c(n);
    for (int i = 0; i < n; ++i) { c[i] = Cand{i, policy[i], 0.0f}; }
    std::sort(c.begin(), c.end(), [](const Cand& l, const Cand& r) { return l.policy > r.policy; });
    for (int i = 0; i < n; ++i) { order[i] = c[i].action; }
}
