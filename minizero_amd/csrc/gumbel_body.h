// Gumbel root logic on the device (ref actor/gumbel_zero.cpp:74-137): sequential halving of the sampled root candidates and the
// choice of the root child the next simulation starts from.  Run by ONE wave64 for game `g` between two simulations of the per-game
// simulation kernel.  Everything here is deterministic given the root statistics (the only random input, the Gumbel noise of the
// root logits, is drawn on the host in the reference's RNG order and applied by simApplyRootNoise); the three std::sort calls of
// the reference are replayed with sort_emul.h, so ties fall the way libstdc++ lets them fall.  The host's own copy of this logic
// (worker.cpp gumbelSequentialHalving / gumbelSortByScore) stays in charge of the move decision: it reads the state back.
#pragma once
#include "gumbel.h"
#include "pool_body.h"
#include "sort_emul.h"

namespace mz {

struct GumbelByLogit { // candidates by logit, descending (gumbel_zero.cpp:97)
    const float* lg;
    __device__ bool operator()(int l, int r) const { return lg[l] > lg[r]; }
};
struct GumbelByScore { // gumbel_zero.cpp:121-137
    const float* score;
    __device__ bool operator()(int l, int r) const { return score[l] > score[r]; }
};
struct GumbelByCount { // fewest visits first, then the larger logit (gumbel_zero.cpp:78-81)
    const float *cnt, *lg;
    __device__ bool operator()(int l, int r) const { return cnt[l] < cnt[r] || (cnt[l] == cnt[r] && lg[l] > lg[r]); }
};

constexpr int kSmallSort = kStdSortInsertionOnly; // libstdc++'s _S_threshold: ranges up to this length are sorted by insertion alone (sort_emul.h)

__device__ __forceinline__ void gumbelWaveSync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// a[0 .. n) (LDS, n <= 64) into the order a STABLE sort by `comp` gives, by all lanes of the wave: element i goes behind the elements that come before it
// under comp and behind the equivalent ones it followed
template <class Comp>
__device__ __forceinline__ void sortSmallStable(int* a, int n, Comp comp, int lane)
{
    n = __builtin_amdgcn_readfirstlane(n);
    const int e = lane < n ? a[lane] : 0;
    const int rank = lane < n ? stableRankOf(a, n, lane, comp) : 0;
    gumbelWaveSync();
    if (lane < n) { a[rank] = e; }
    gumbelWaveSync();
}

// the step by one lane, every std::sort replayed with sort_emul.h (any number of candidates)
__device__ __forceinline__ int gumbelStepSerial(const PoolView& v, const GumbelView& gv, int sim_post, int nc, int fc, float mx, float* cnt, float* lg, float* score,
                                                int* idx, int* cand, int* stack, int* st, int bump)
{
    int start = 0;
    {
        int ncand = st[0], sample = st[1], budget = st[2];
        bool defer = false;
        if (sim_post != 1) { for (int i = 0; i < ncand; ++i) { cand[i] = st[3 + i]; } }
        if (sim_post == 1) { // the root has just been expanded: sample the top-m children by (noisy) logit
            StdSortEmul<int, GumbelByLogit> s{idx, GumbelByLogit{lg}};
            s.sort(nc, stack);
            ncand = nc < gv.sample_size ? nc : gv.sample_size;
            for (int i = 0; i < ncand; ++i) { cand[i] = idx[i]; }
            sample = gv.sample_size;
            budget = gv.budget0;
        } else {
            bool all = true;
            for (int i = 0; i < ncand; ++i) { if (!(cnt[cand[i]] >= static_cast<float>(budget))) { all = false; break; } }
            if (all && bump >= 0) { defer = true; }
            if (all && bump < 0) {
                // int / (double * int / int): three IEEE double operations, no contraction (the reference's promotions)
                const double nb = __builtin_floor(static_cast<double>(gv.num_simulation) / (gv.log2_m * static_cast<double>(sample) / 2.0));
                const int next_budget = nb >= 2147483647.0 ? 2147483647 : static_cast<int>(nb);
                if (next_budget > 0 && sample > 2) {
                    sample /= 2;
                    for (int i = 0; i < nc; ++i) {
                        const float value = score[i];
                        const float sc = lg[i] + (gv.sigma_visit_c + mx) * gv.sigma_scale_c * value;
                        score[i] = cnt[i] > 0.0f ? sc : -3.402823466e+38f;
                    }
                    StdSortEmul<int, GumbelByScore> s{cand, GumbelByScore{score}};
                    s.sort(ncand, stack);
                    if (ncand > sample) { ncand = sample; }
                    budget = static_cast<int>(cnt[cand[0]] + static_cast<float>(next_budget));
                }
            }
        }
        if (defer) {
            start = -1;
        } else {
            StdSortEmul<int, GumbelByCount> s2{cand, GumbelByCount{cnt, lg}};
            s2.sort(ncand, stack);
            st[0] = ncand; st[1] = sample; st[2] = budget;
            for (int i = 0; i < ncand; ++i) { st[3 + i] = cand[i]; }
            start = fc + cand[0];
        }
    }
    return start;
}

inline size_t gumbelSmemBytes(int A) { return size_t(A) * (3 * sizeof(float) + sizeof(int)) + (kGumbelMaxSample + 3 * 40) * sizeof(int) + 64; }

// sim_post = simulations of this search completed so far (>= 1).  Returns the node the next selection starts from.
// bump >= 0: the step is computed AHEAD of the backup of the simulation in flight — that backup will add one visit to root child `bump` (the child on the
// current path) and to nothing else this step reads, unless the candidates have all reached their budget (the halving ranks them by their means, which
// the backup changes): then nothing is written and -1 comes back, and the step runs again after the backup (sim_cluster.h).
// bump_cnt >= 0: the visit count of child `bump` BEFORE that backup, read by the caller while no backup was in flight (the step then runs beside the
// backup on another wave, sim_kernel_mz: whatever the record of that child holds at the moment is not looked at)
// kids != nullptr (only with bump >= 0): the caller keeps the root's first_child / num_children (kids[0], kids[1]) and its children's visit counts and logits
// (kids + 2, kids + 2 + A: floats) up to date in LDS — the counts without the backup in flight — so the step ahead of a backup makes no trip to global memory: it
// never looks at the children's means (a step that would, the halving, returns -1 first)
__device__ __forceinline__ int gumbelStepBody(const PoolView& v, const GumbelView& gv, int sim_post, int g, int lane, float* smem, int bump = -1, float bump_cnt = -1.0f,
                                              const int* kids = nullptr)
{
    const size_t base = size_t(g) * v.cap;
    const bool from_kids = kids != nullptr && bump >= 0 && sim_post != 1;
    NodeRec root;
    if (from_kids) { root.first_child = kids[0]; root.num_children = kids[1]; root.players = 0; }
    else { root = v.rec[base]; }
    const int nc = root.num_children, fc = root.first_child, cplayer = (root.players >> 8) & 0xFF;
    float* cnt = smem;              // [nc]
    float* lg = cnt + v.A;          // [nc]
    float* score = lg + v.A;        // [nc]
    int* idx = reinterpret_cast<int*>(score + v.A); // [nc] scratch for the first sort
    int* cand = idx + v.A;                           // [kGumbelMaxSample] the candidates, sorted in LDS
    int* stack = cand + kGumbelMaxSample;            // sort_emul range stack
    int* st = gv.state + size_t(g) * (3 + kGumbelMaxSample);
    float mx = 0.0f;
    if (from_kids) {
        const float* kc = reinterpret_cast<const float*>(kids + 2);
        for (int i = lane; i < nc; i += 64) {
            cnt[i] = i == bump ? kc[i] + 1.0f : kc[i];
            lg[i] = kc[v.A + i];
            score[i] = 0.0f;
            idx[i] = i;
        }
    }
    const int bsize = from_kids ? 0 : v.bound_size[g];
    const float lo = from_kids ? 0.0f : v.bound_lo[g], hi = from_kids ? 0.0f : v.bound_hi[g];
    for (int i = lane; i < nc && !from_kids; i += 64) {
        const NodeRec c = v.rec[base + fc + i];
        cnt[i] = i == bump ? (bump_cnt >= 0.0f ? bump_cnt : c.count) + 1.0f : c.count;
        lg[i] = v.logit[base + fc + i];
        mx = c.count > mx ? c.count : mx;
        // score of gumbelSortByScore: logit + (c_visit + max count) * c_scale * normalized mean; the max count is added below
        score[i] = c.count > 0.0f ? normalizedMean(v, c.reward, c.mean, c.count, cplayer, bsize, lo, hi) : 0.0f;
        idx[i] = i;
    }
    for (int o = 32; o > 0; o >>= 1) { const float m2 = __shfl_xor(mx, o); mx = m2 > mx ? m2 : mx; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int start = 0;
    if (sim_post == 1 || st[0] > kSmallSort) { // once per move (the root's children, more than 16: libstdc++'s introsort replayed by one lane) / sample sizes beyond 16
        if (lane == 0) { start = gumbelStepSerial(v, gv, sim_post, nc, fc, mx, cnt, lg, score, idx, cand, stack, st, bump); }
        return __builtin_amdgcn_readfirstlane(start);
    }
    // <= 16 candidates: std::sort is libstdc++'s plain insertion sort there (bits/stl_algo.h __final_insertion_sort with n <= _S_threshold), i.e. a STABLE
    // sort — its result is the one permutation a rank count gives: all lanes at once instead of ~100 dependent comparisons of one lane (6.5 -> ~1.5 us
    // per simulation; tests/test_sort_emul.py checks sortSmallStable against the real std::sort on tie-heavy inputs)
    int ncand = __builtin_amdgcn_readfirstlane(st[0]), sample = __builtin_amdgcn_readfirstlane(st[1]), budget = __builtin_amdgcn_readfirstlane(st[2]);
    if (lane < ncand) { cand[lane] = st[3 + lane]; }
    gumbelWaveSync();
    const bool reached = lane < ncand ? cnt[cand[lane]] >= static_cast<float>(budget) : true;
    const bool all = __ballot(!reached) == 0;
    if (all && bump >= 0) { return -1; } // the halving ranks the candidates by means the backup in flight still changes: again after the backup
    if (all) {
        // int / (double * int / int): three IEEE double operations, no contraction (the reference's promotions)
        const double nb = __builtin_floor(static_cast<double>(gv.num_simulation) / (gv.log2_m * static_cast<double>(sample) / 2.0));
        const int next_budget = nb >= 2147483647.0 ? 2147483647 : static_cast<int>(nb);
        if (next_budget > 0 && sample > 2) {
            sample /= 2;
            for (int i = lane; i < nc; i += 64) {
                const float value = score[i];
                const float sc = lg[i] + (gv.sigma_visit_c + mx) * gv.sigma_scale_c * value;
                score[i] = cnt[i] > 0.0f ? sc : -3.402823466e+38f;
            }
            gumbelWaveSync();
            sortSmallStable(cand, ncand, GumbelByScore{score}, lane);
            if (ncand > sample) { ncand = sample; }
            budget = static_cast<int>(cnt[cand[0]] + static_cast<float>(next_budget));
        }
    }
    sortSmallStable(cand, ncand, GumbelByCount{cnt, lg}, lane);
    if (lane == 0) { st[0] = ncand; st[1] = sample; st[2] = budget; }
    if (lane < ncand) { st[3 + lane] = cand[lane]; }
    start = fc + cand[0];
    return __builtin_amdgcn_readfirstlane(start);
}

} // namespace mz
