// The device functions the MuZero per-game simulation kernels share (sim.hip sim_kernel_mz / sim_pre_kernel_mz / the cluster kernel, sim_wide_mz.hip sim_kernel_mz_wide):
// walk, probe of a leaf evaluated ahead, candidate list, expand + backup, hidden-state rescale, heads.
#pragma once
#include "sim_az_body.h"

namespace mz {

// ---- MuZero (board games; ref muzero_network.h:97-178, zero_actor.cpp:215-245): no leaf environment.  The leaf is evaluated from
// its parent's hidden state (slab slot `hslot[parent]`) and the move; its children are ALL actions (the root: the legal ones) in
// the reference's sort order; the new hidden state is rescaled to [0, 1] per sample and written to the slab slot of this simulation.
__device__ __noinline__ void simMzCandGather(CSimArgs* __restrict__ a, int g, int lane, float* tiles, int* kshare, int* lds_path = nullptr)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
    const PoolView v = lds_path ? simPathView(ldc(&a->pv), lds_path, g) : ldc(&a->pv);
    const int A = a->A, len = v.path_len[g], depth = len - 1;
    Cand* cs = reinterpret_cast<Cand*>(tiles);
    int k = 0;
    for (int base = 0; base < A; base += 64) {
        const int ac = base + lane;
        const bool leg = ac < A && (depth > 0 || ((a->root_legal[size_t(g) * a->LW + (ac >> 6)] >> (ac & 63)) & 1)); // legality is only known at the root (zero_actor.cpp:238)
        const unsigned long long m = __ballot(leg);
        if (leg) { cs[k + __popcll(m & ((1ull << lane) - 1))] = Cand{ac, a->policy[size_t(g) * A + ac], a->logit[size_t(g) * A + ac]}; }
        k += __popcll(m);
    }
    waveSync();
    if (k > kCandCoopMax || !a->cand_coop) { orderCandidates(cs, cs + A, reinterpret_cast<int*>(cs + 2 * A), k, lane, a->err); }
    else { candDense(cs, k, lane, simCandDense(tiles, A)); }
    if (lane == 0) { *kshare = k; }
}

// given: the network outputs of this leaf come from the stand-alone kernels (the muzero_atari root: value / reward still in the transformed scale)
// part 0: everything; part 1: the candidate list and the new children (needs the policy only); part 2: value / reward + backup (the cluster kernel runs
// part 1 while the value and reward heads of the game's other workgroups are still busy)
// presorted: a leaf that was evaluated ahead (simPreProbe) — `slot` is its entry, which holds the sorted candidate list
// lds_path: the simulation's path lives in LDS (simPathView) instead of the pool's arrays
__device__ __noinline__ void simMzCandExpand(CSimArgs* __restrict__ a, int slot, int g, int lane, float* tiles, int k, bool given = false, int part = 0, bool presorted = false,
                                             int* lds_path = nullptr)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
    slot = __builtin_amdgcn_readfirstlane(slot);
    k = __builtin_amdgcn_readfirstlane(k);
    part = __builtin_amdgcn_readfirstlane(part);
    const PoolView v = lds_path ? simPathView(ldc(&a->pv), lds_path, g) : ldc(&a->pv);
    const int A = a->A, len = v.path_len[g], depth = len - 1;
    if (part != 2 && !presorted) {
        Cand* cs = reinterpret_cast<Cand*>(tiles);
        Cand* out = cs + A;
        if (k > 0 && k <= kCandCoopMax && a->cand_coop) {
            float* dense = simCandDense(tiles, A);
            candScatter(cs, out, reinterpret_cast<int*>(out + A), k, 8, lane, reinterpret_cast<const int*>(dense + kCandCoopMax), a->err);
        }
        for (int i = lane; i < k; i += 64) {
            a->cand_action[size_t(g) * A + i] = out[i].action;
            a->cand_policy[size_t(g) * A + i] = out[i].policy;
            a->cand_logit[size_t(g) * A + i] = out[i].logit;
        }
    }
    if (presorted && part == 0) {
        // A leaf evaluated ahead: its sorted candidate list lies in its entry (= its slab slot) — all A actions (legality is only known at the root,
        // zero_actor.cpp:238) — and its value and reward too; everything expand + backup need is loaded side by side, nothing goes through the game's arrays
        const size_t e = size_t(g) * a->slots + slot, off = (e - g) * A;
        const int rt = a->root_turn[g];
        const ExpandGiven eg{k, (a->num_players == 2 && (depth & 1)) ? 3 - rt : rt, a->pre_value[e], a->atari ? a->pre_reward[e] : 0.0f};
        MZ_LPROF(23);
        expandBackupBody(v, a->cand_count, a->pre_action + off, a->pre_policy + off, a->pre_logit + off, a->cand_player, a->value_io, a->reward_io, slot, a->err, g, lane, tiles, 0, &eg);
        MZ_LPROF(24);
        return;
    }
    if (lane == 0) {
        if (part != 2) {
            const int rt = a->root_turn[g];
            a->cand_count[g] = k;
            a->cand_player[g] = (a->num_players == 2 && (depth & 1)) ? 3 - rt : rt; // the children are moved by the player to move at the leaf
        }
        if (part != 1) {
            const bool inv = given && a->atari; // the host path's invertValueHost() of both (worker.cpp buildCandidates)
            a->value_io[g] = inv ? invertValueDev(a->value[g]) : a->value[g];
            a->reward_io[g] = a->atari ? (inv ? invertValueDev(a->reward[g]) : a->reward[g]) : 0.0f; // board games have no reward head (ref muzero_network.h:129)
        }
    }
    waveSync();
    MZ_LPROF(23);
    if (presorted) { // (cluster kernel, part 1: the children of a leaf evaluated ahead; part 2 takes value and reward from the game's arrays, simPreProbe put them there)
        const size_t off = (size_t(g) * a->slots + slot - g) * A;
        expandBackupBody(v, a->cand_count, a->pre_action + off, a->pre_policy + off, a->pre_logit + off, a->cand_player, a->value_io, a->reward_io, slot, a->err, g, lane, tiles, part);
    } else {
        expandBackupBody(v, a->cand_count, a->cand_action, a->cand_policy, a->cand_logit, a->cand_player, a->value_io, a->reward_io, slot, a->err, g, lane, tiles, part);
    }
    MZ_LPROF(24);
}

// The Gumbel step of simulation `next_slot`, ahead of the backup of the simulation in flight (gumbel_body.h `bump`): true if a->start[g] and the state are
// those the step after the backup would write.  The cluster kernel runs it on its owner while the other workgroups' value / reward heads are still busy.
__device__ __noinline__ bool simGumbelAhead(CSimArgs* __restrict__ a, int next_slot, int g, int lane, float* tiles, float bump_cnt = -1.0f, int* lds_path = nullptr,
                                            int* state_lds = nullptr, const int* kids = nullptr)
{
    g = __builtin_amdgcn_readfirstlane(g);
    next_slot = __builtin_amdgcn_readfirstlane(next_slot);
    const PoolView pv = lds_path ? simPathView(ldc(&a->pv), lds_path, g) : ldc(&a->pv);
    GumbelView gum = ldc(&a->gum);
    if (state_lds) { gum.state = state_lds - size_t(g) * (3 + kGumbelMaxSample); }
    const int len = pv.path_len[g];
    if (len < 2) { return false; }
    const int child = pv.path[size_t(g) * pv.max_depth + 1] - (kids ? kids[0] : pv.rec[size_t(g) * pv.cap].first_child);
    const int st = gumbelStepBody(pv, gum, next_slot, g, lane, tiles, child, bump_cnt, kids);
    if (st >= 0 && lane == 0) { a->start[g] = st; }
    waveSync();
    return st >= 0;
}

__device__ __noinline__ void simMzSelect(CSimArgs* __restrict__ a, int slot, bool host_start, int g, int lane, float* tiles, LdsCDouble* rcp, SpecMem spec,
                                         bool gumbel_done = false, bool noise_done = false, int serial = 0, int* lds_path = nullptr, int* state_lds = nullptr)
{
    serial = __builtin_amdgcn_readfirstlane(serial);
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    slot = __builtin_amdgcn_readfirstlane(slot);
    g = __builtin_amdgcn_readfirstlane(g);
    MZ_LPROF(0);
    if (slot == 1 && a->root_noise && !noise_done) { simApplyRootNoise<2>(a, g, lane); }
    if (a->use_gumbel && !gumbel_done) { simGumbelStart<2>(a, slot, host_start, g, lane, tiles, state_lds); }
    MZ_LPROF(20);
    const PoolView pv = lds_path ? simPathView(ldc(&a->pv), lds_path, g) : ldc(&a->pv);
    selectBody<true>(pv, a->use_gumbel ? a->start : nullptr, g, lane, rcp, spec, serial);
    MZ_LPROF(21);
}

// Leaves evaluated ahead (sim_pre_kernel_mz): does the entry of simulation `slot` hold THIS leaf — the same parent hidden state (slab slot `src`) and the
// same action, written in this move (`epoch`)?  Then its value and reward are copied to the game's arrays and the slab slot
// that holds its hidden state is returned (`slot`, or alt_base + slot for the second expected leaf); -1: not evaluated ahead.  The simulation then runs
// exactly as if the kernel had evaluated the leaf itself: candidates, expand (the node remembers the returned slot), backup.
__device__ __forceinline__ int simPreProbe(CSimArgs* __restrict__ a, int epoch, int g, int slot, int src, int action, int lane)
{
    if (epoch == 0 || !a->pre_key) { return -1; }
    size_t e = size_t(g) * a->slots + slot;
    const int* key = a->pre_key + e * 4;
    bool hit = __builtin_amdgcn_readfirstlane((key[2] == epoch && key[0] == src && key[1] == action) ? 1 : 0) != 0;
    int eslot = slot;
    if (!hit && a->alt_base) { // the second expected leaf of this simulation (entry and slab slot alt_base + slot)
        const int* key2 = key + size_t(a->alt_base) * 4;
        hit = __builtin_amdgcn_readfirstlane((key2[2] == epoch && key2[0] == src && key2[1] == action) ? 1 : 0) != 0;
        if (hit) { e += a->alt_base; eslot += a->alt_base; if (a->pre_stat && lane == 0) { atomicAdd(a->pre_stat + 128, 1u); if (slot < 126) { atomicAdd(a->pre_stat + 256 + slot, 1u); } } } // ([256 + s]: by simulation, Worker::adaptRounds)
    }
    if (!hit) {
        if (a->pre_stat && key[2] == epoch && lane == 0 && slot < 126) { atomicAdd(a->pre_stat + 2 + slot, 1u); } // (monitoring: which simulations of a move miss, MZ_SIM_PROF)
        return -1;
    }
    if (lane == 0) { // (the entry's sorted candidate list is read where it lies: simMzCandExpand)
        a->value[g] = a->pre_value[e];
        a->reward[g] = a->pre_reward[e];
        if (a->pre_stat) { atomicAdd(a->pre_stat, 1u); }
    }
    waveSync();
    return eslot;
}

// scale_hidden_state (ref muzero_network.py:81-88) of the tower's output where it lies (padded planes in LDS), in place, and the rescaled state
// to the slab slot `hd`: min / max are order-free, (h - min) / scale is one IEEE operation per element.  H, W compile-time: with run-time
// geometry the three integer divisions per element and pass cost more than the arithmetic (heads 21.5 -> 13 us on BASELINE configs[3])
template <int H, int W, int CS = planeStride(H, W)>
__device__ __forceinline__ void rescaleTile(float* __restrict__ xt, int C, float* __restrict__ hd, int tid, float* __restrict__ red)
{
    constexpr int P = H * W, PW = W + 2;
    const int lane = tid & 63, wave = tid >> 6;
    MZ_HPROF(0);
    float mn = 3.4e38f, mx = -3.4e38f;
    for (int i = tid; i < C * P; i += 512) {
        const int c = i / P, p = i - c * P;
        const float v = xt[c * CS + (p / W + 1) * PW + p % W + 1];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mn, o), x2 = __shfl_xor(mx, o);
        mn = m2 < mn ? m2 : mn;
        mx = x2 > mx ? x2 : mx;
    }
    if (lane == 0) { red[wave] = mn; red[8 + wave] = mx; }
    __syncthreads();
    MZ_HPROF(1);
    mn = red[0]; mx = red[8];
    for (int w = 1; w < 8; ++w) { mn = red[w] < mn ? red[w] : mn; mx = red[8 + w] > mx ? red[8 + w] : mx; }
    float scale = mx - mn;
    if (scale < 1e-5f) { scale += 1e-5f; }
    for (int i = tid; i < C * P; i += 512) {
        const int c = i / P, p = i - c * P, k = c * CS + (p / W + 1) * PW + p % W + 1;
        const float v = (xt[k] - mn) / scale;
        xt[k] = v;
        hd[i] = v;
    }
    MZ_HPROF(2);
    __syncthreads();
    MZ_HPROF(3);
}

template <int H, int W>
__device__ __forceinline__ void simMzHeads(CSimArgs* __restrict__ a, int slot, int g, int tid, float* tiles, float* scratch, float* xtile)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    slot = __builtin_amdgcn_readfirstlane(slot);
    g = __builtin_amdgcn_readfirstlane(g);
    constexpr int xcs = planeStride(H, W), xpw = W + 2;
    if (a->atari) { // 601-bin value / reward heads, rescaled hidden state to the slab slot of this simulation, value and reward in game scale
        const AtariHeadParams hp = ldc(&a->ahp);
        float* hd = a->hidden + (size_t(g) * a->slots + slot) * size_t(hp.C) * hp.P;
        atariHeadsBody<256>(nullptr, xtile, xcs, xpw, hp, a->policy, a->logit, a->value, a->reward, hd, 1, 1, g, tid, scratch);
        return;
    }
    const HeadParams hp = ldc(&a->hp);
    float* hd = a->hidden + (size_t(g) * a->slots + slot) * size_t(hp.C) * hp.P;
    rescaleTile<H, W>(xtile, hp.C, hd, tid, tiles); // tile 0 (the blocks' temporary) is free: its first words hold the reduction scratch
    headsBody(nullptr, hp, a->policy, a->logit, a->value, nullptr, nullptr, 0, g, tid, 512, tiles, xtile, xcs, xpw);
    MZ_HPROF(4);
}

} // namespace mz
