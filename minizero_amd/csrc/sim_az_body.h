// The device side of the AlphaZero per-game simulation kernel (sim_kernel and the phase functions it shares with the MuZero kernels of sim.hip), as a header:
// sim.hip instantiates the BASELINE shapes (two-tile tower), sim_wide.hip the wide / large-board shapes (one-tile tower, net_wide_body.h).
#pragma once
// sim_kernel — the per-game simulation kernel: workgroup g runs `nsims` complete MCTS simulations of game g without leaving
// the GPU: PUCT selection, the leaf's Go position / planes / legal mask, the residual tower + heads on the 8 waves of the
// workgroup, the candidate list and expand + backup.  The reference steps all games in lock-step, one batched forward per
// cycle (ref actor/actor_group.cpp:81-114); nothing in a game depends on another game, so here every game advances at its own
// pace: a simulation costs ITS path depth, not the deepest of the 256 paths, there are no kernel boundaries inside a move, and
// the tower of one game overlaps the tree phases of the others.  The per-sample arithmetic is that of the stand-alone kernels
// (same device bodies: net_body.h, pool_body.h, go_body.h), so results are bit-identical to the lock-step path.
// The host draws the per-cycle feature rotations in the reference's order (cycle-major, actor-minor) before the launch.
#include "net.h"
#ifdef MZ_SIM_TPROF // experiment: cycles per tower layer inside the simulation kernel (game 0, wave 0), printed by dumpSimProf
__device__ unsigned long long g_tp[64];
__shared__ unsigned long long s_tp_prev;
__shared__ int s_tp_idx;
#define MZ_TPROF(slot)                                                                         \
    do {                                                                                       \
        if ((slot) == 3 && threadIdx.x == 0 && blockIdx.x == 0) {                              \
            const unsigned long long t_ = clock64();                                           \
            g_tp[s_tp_idx & 63] += t_ - s_tp_prev; s_tp_prev = t_; ++s_tp_idx;                 \
        }                                                                                      \
    } while (0)
#endif
#ifdef MZ_SIM_HPROF // experiment: where the time of the in-kernel 601-bin heads goes (game 0)
#ifndef MZ_HPROF_BLOCK
#define MZ_HPROF_BLOCK 0 // cluster mode: 64 = member 1 (reward head) of game 0 in a pool of 64 games
#endif
__device__ unsigned long long g_hp[16];
__shared__ unsigned long long s_hp_prev;
#define MZ_HPROF(k)                                                                                   \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x == MZ_HPROF_BLOCK) {                                       \
            const unsigned long long t_ = wall_clock64();                                             \
            if ((k) > 0) { g_hp[(k)] += t_ - s_hp_prev; } else { g_hp[15] += 1; }                     \
            s_hp_prev = t_;                                                                           \
        }                                                                                             \
    } while (0)
#endif
#include "net_body.h"
#include "net_bf16_body.h"
#include "net_atari_body.h"
#ifdef MZ_SIM_LPROF // experiment: where the time of the single-wave tree phases goes (game 0)
__device__ unsigned long long g_lp[32];
__shared__ unsigned long long s_lp_prev;
#define MZ_LPROF(k)                                                                                   \
    do {                                                                                              \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {                                             \
            const unsigned long long t_ = wall_clock64();                                             \
            if ((k) > 0) { g_lp[(k)] += t_ - s_lp_prev; } else { g_lp[31] += 1; }                     \
            s_lp_prev = t_;                                                                           \
        }                                                                                             \
    } while (0)
#endif
#ifdef MZ_SIM_BPROF // experiment: the part of the Go leaf that runs beside the heads (game 0): [role][0] entry -> start, [1] first piece, [2] wait at barrier 1, [3] legal mask, [4] wait at barrier 2
__device__ unsigned long long g_bp[20];
#define MZ_BPROF(role, k)                                                                              \
    do {                                                                                               \
        if (PART == 2 && (threadIdx.x & 63) == 0 && blockIdx.x == 0) {                                 \
            const unsigned long long t_ = wall_clock64();                                              \
            if ((k) == 0) { g_bp[16 + (role)] += 1; g_bp[18 + (role)] = t_; }                          \
            else { g_bp[(role) * 8 + (k)] += t_ - g_bp[18 + (role)]; g_bp[18 + (role)] = t_; }        \
        }                                                                                              \
    } while (0)
#endif
#include "pool_body.h"
#include "go_body.h"
#include "gumbel_body.h"
#include "sim_args.h"
#include <algorithm>
#include <type_traits>
#include <cstring>
#include <cstdlib>
#include <vector>

#ifndef MZ_HEADS_FP
#define MZ_HEADS_FP 0 // (experiment switch: the two-tile 9x9 kernels' heads with global / DS loads instead of flat loads)
#endif

namespace mz {


// The heads' outputs and the candidate list of a simulation never leave the CU: they are handed from phase to phase through a small LDS
// block instead of global memory (each hand-over was a store + a dependent load through L2).  The bodies index their arrays with the
// game index, so they get generic pointers moved back by the game's offset.
struct SimXchg { // word offsets inside the block for A actions
    int A;
    __device__ int policy() const { return 0; }
    __device__ int logit() const { return A; }
    __device__ int cpolicy() const { return 2 * A; }
    __device__ int clogit() const { return 3 * A; }
    __device__ int caction() const { return 4 * A; }
    __device__ int scalars() const { return 5 * A; } // value, cand_count, cand_player, value_io, reward_io, leaf_player, terminal, eval, + the backup wave's value / reward
    __device__ int legal() const { return 5 * A + 12; }   // 64-bit words of the leaf's legal mask (8-byte aligned: A is padded to even below)
    __device__ int feat() const { return 5 * A + 12 + 16; } // the leaf's bit-packed planes (the tower's input)
};
inline size_t simXchgWords(int A, int channels, int W32) { return size_t(5) * (A + (A & 1)) + 12 + 16 + size_t(channels) * W32; }
__device__ __forceinline__ int simXchgWordsDev(int A, int channels, int W32) { return 5 * (A + (A & 1)) + 12 + 16 + channels * W32; }


// the leaf's outputs (planes, legal mask, player, terminal flag, result) go to the next phases through the hand-over block too
__device__ __forceinline__ GoDevView simLeafView(GoDevView gv, float* xchg, int g)
{
    const SimXchg x{gv.A + (gv.A & 1)};
    float* sc = xchg + x.scalars();
    gv.leaf_player = reinterpret_cast<int*>(sc + 5) - g;
    gv.terminal = reinterpret_cast<int*>(sc + 6) - g;
    gv.eval = sc + 7 - g;
    gv.legal = reinterpret_cast<uint64_t*>(xchg + x.legal()) - size_t(g) * gv.LW;
    gv.feat = reinterpret_cast<uint32_t*>(xchg + x.feat()) - size_t(g) * gv.channels * gv.W32;
    return gv;
}

// The tree phases are separate (non-inlined) functions: inlined next to the tower they push the kernel to 256 VGPRs with spills in
// the MFMA loop.  SimArgs lives in device memory (not in 1.3 KB of kernel arguments pinned in SGPRs for the whole kernel).
typedef __attribute__((address_space(3))) const double LdsCDouble;

template <int CPL, int WPE, class RcpPtr>
__device__ __noinline__ void simSelectLeaf(CSimArgs* __restrict__ a, int rot, int slot, int g, int lane, float* tiles, RcpPtr rcp, SpecMem spec,
                                           float* xchg, const uint64_t* seen_lds, int serial = 0, uint64_t* leaf_smem = nullptr)
{
    serial = __builtin_amdgcn_readfirstlane(serial);
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
    slot = __builtin_amdgcn_readfirstlane(slot);
    rot = __builtin_amdgcn_readfirstlane(rot);
    unsigned long long t0 = 0;
    if (a->prof) { t0 = wall_clock64(); }
    const PoolView pv = simPathView(ldc(&a->pv), reinterpret_cast<int*>(xchg) - 2 * a->pv.max_depth - 2, g);
#ifdef MZ_SELECT_TWICE // experiment: the walk again, now with its records in the caches -> the profile shows the arithmetic-only time
    selectBody<WPE == 2>(pv, a->use_gumbel ? a->start : nullptr, g, lane, rcp, spec);
    waveSync();
    if (a->prof) { t0 = wall_clock64(); }
#endif
    selectBody<WPE == 2>(pv, a->use_gumbel ? a->start : nullptr, g, lane, rcp, spec, serial);
    waveSync();
    if (a->prof && lane == 0) {
        a->prof[size_t(g) * 8 + 5] += wall_clock64() - t0;
        a->prof[size_t(g) * 8 + 6] += pv.path_len[g];
    }
    MZ_LPROF(0);
    const GoDevView gv = simLeafView(ldc(&a->gv), xchg, g);
    if constexpr (CPL == -1) { tttLeafBody(gv, pv, rot, slot, g, lane); } // CPL -1: TicTacToe, 0: Othello (go_body.h)
    else if constexpr (CPL == 0) { othLeafBody(gv, pv, rot, slot, g, lane); }
    else if (leaf_smem) { goLeafBody<CPL, true, 1>(gv, pv, rot, slot, g, lane, leaf_smem, seen_lds); } // what the network needs; the rest beside the heads (simLeafRest)
    else { goLeafBody<CPL, true>(gv, pv, rot, slot, g, lane, reinterpret_cast<uint64_t*>(tiles), seen_lds); } // planes: simLeafPlanes, all waves
}

// Go, one game per CU: what only the phases after the network need of the leaf — path hashes, liberties, legal mask, a terminal leaf's score (go_body.h
// goLeafBody PART 2) — on the workgroup's last two waves BESIDE the heads, in which those waves have no share.  They pass the two barriers headsBody passes.
template <int CPL>
__device__ __noinline__ void simLeafRest(CSimArgs* __restrict__ a, int rot, int slot, int g, int lane, float* xchg, const uint64_t* seen_lds, uint64_t* leaf_smem, int role)
{
    role = __builtin_amdgcn_readfirstlane(role);
    g = __builtin_amdgcn_readfirstlane(g);
    slot = __builtin_amdgcn_readfirstlane(slot);
    rot = __builtin_amdgcn_readfirstlane(rot);
    if constexpr (CPL > 0) {
        const PoolView pv = simPathView(ldc(&a->pv), reinterpret_cast<int*>(xchg) - 2 * a->pv.max_depth - 2, g);
        const GoDevView gv = simLeafView(ldc(&a->gv), xchg, g);
        if (role == 0) { goLeafBody<CPL, true, 2, true, 0>(gv, pv, rot, slot, g, lane, leaf_smem, seen_lds); }
        else { goLeafBody<CPL, true, 2, true, 1>(gv, pv, rot, slot, g, lane, leaf_smem, seen_lds); }
    }
}

// waves 1 .. 3 beside wave 0's walk: levels 17 .. 64 of the path the previous simulation took, 16 per wave (pool_body.h selectSpecHelper): a deep principal variation is
// re-walked by almost every simulation, and the launch lasts as long as its deepest game
template <class RcpPtr>
__device__ __noinline__ void simSelectHelper(CSimArgs* __restrict__ a, int g, int lane, int seg, int serial, RcpPtr rcp, SpecMem spec)
{
    g = __builtin_amdgcn_readfirstlane(g);
    seg = __builtin_amdgcn_readfirstlane(seg);
    serial = __builtin_amdgcn_readfirstlane(serial);
    const PoolView pv = ldc(&a->pv);
    selectSpecHelper(pv, g, lane, seg, serial, rcp, spec);
}

// Go: the 18 feature planes of the leaf, two or three per wave (32 ballots over LDS words: 3.3 us on one wave)
template <int CPL>
__device__ __forceinline__ void simLeafPlanes(CSimArgs* __restrict__ a, int rot, int g, int wave, int lane, const uint64_t* leaf_smem, float* xchg)
{
    if constexpr (CPL > 0) {
        const GoDevView gv = simLeafView(ldc(&a->gv), xchg, g);
        goPlanesPart<CPL>(gv, a->pv.max_depth, rot, g, wave, 8, lane, leaf_smem);
    }
}

// Candidates + expand + backup in three steps: wave 0 gathers the legal actions, ALL waves count ranks (the sort is VALU-bound and the other
// seven waves would be idle), wave 0 scatters, expands and backs up.  A > 128 actions: wave 0 sorts alone in the first step.
__device__ __forceinline__ float* simCandDense(float* tiles, int A) { return reinterpret_cast<float*>(reinterpret_cast<char*>(tiles) + ((2 * size_t(A) * sizeof(Cand) + kSortStackBytes + 16 + 15) & ~size_t(15))); }

template <int WPE>
__device__ __forceinline__ void simCandGatherImpl(CSimArgs* __restrict__ a, int rot, int g, int lane, float* tiles, float* xchg)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
    rot = __builtin_amdgcn_readfirstlane(rot);
    const GoDevView gv = simLeafView(ldc(&a->gv), xchg, g);
    const SimXchg x{gv.A + (gv.A & 1)};
    const size_t ga = size_t(g) * gv.A;
    float* sc = xchg + x.scalars();
    MZ_LPROF(6);
    Cand* cs = reinterpret_cast<Cand*>(tiles);
    const int k = azCandGather(gv, xchg + x.policy() - ga, xchg + x.logit() - ga, rot, g, lane, cs);
    waveSync();
    // (cand_coop 2: boards of more than 128 actions — up to six candidates per lane, candRankPartW)
    if (a->cand_coop == 2 && k > 0 && k <= kCandCoopMaxW) { candDenseW(cs, k, lane, simCandDense(tiles, gv.A)); }
    else if (k > kCandCoopMax || !a->cand_coop) { if (k > 0) { orderCandidates(cs, cs + gv.A, reinterpret_cast<int*>(cs + 2 * gv.A), k, lane, a->err); } }
    else { candDense(cs, k, lane, simCandDense(tiles, gv.A)); }
    if (lane == 0) { reinterpret_cast<int*>(sc)[1] = k; } // cand_count: read by every wave after the barrier
    MZ_LPROF(7);
}

// (a function of its own with its own register budget.  It saves and restores the callee-saved VGPRs it uses — 25 here, 78 in simCandExpand, 256 bytes each per call, one
// wave — which is what is left of BASELINE configs[2]'s HBM traffic; calling the bodies inline in the 128-VGPR kernels was measured and lost: the kernel then spills 47
// VGPRs whose reloads sit in these single-wave phases — 32.5 -> 47.8 MB per cycle, 2.03 -> 2.02 M leaf-evals/s)
template <int WPE>
__device__ __noinline__ void simCandGather(CSimArgs* __restrict__ a, int rot, int g, int lane, float* tiles, float* xchg)
{
    simCandGatherImpl<WPE>(a, rot, g, lane, tiles, xchg);
}

__device__ __forceinline__ void simCandRank(int A, int k, int wave, int lane, float* tiles, int coop = 1)
{
    if (coop == 2) {
        if (k > 0 && k <= kCandCoopMaxW) { float* dense = simCandDense(tiles, A); candRankPartW(dense, k, wave, 8, lane, reinterpret_cast<int*>(dense + kCandCoopMaxW)); }
        return;
    }
    if (k <= 0 || k > kCandCoopMax) { return; }
    float* dense = simCandDense(tiles, A);
    candRankPart(dense, k, wave, 8, lane, reinterpret_cast<int*>(dense + kCandCoopMax));
}

// The backup of a simulation on its own wave (wave 1) beside wave 0's scatter + expand: it only needs the leaf's value (the heads' output, or
// the game result at a terminal leaf: zero_actor.cpp:85) and the path.  Not with value rescaling (its multiset shares the scratch).
template <int WPE>
__device__ __noinline__ void simBackupOnly(CSimArgs* __restrict__ a, int slot, int g, int lane, float* tiles, float* xchg)
{
    g = __builtin_amdgcn_readfirstlane(g);
    slot = __builtin_amdgcn_readfirstlane(slot);
    const GoDevView gv = simLeafView(ldc(&a->gv), xchg, g);
    const PoolView pv = simPathView(ldc(&a->pv), reinterpret_cast<int*>(xchg) - 2 * a->pv.max_depth - 2, g);
    const SimXchg x{gv.A + (gv.A & 1)};
    float* sc = xchg + x.scalars();
    if (lane == 0) {
        sc[8] = gv.terminal[g] != 0 ? gv.eval[g] : sc[0];
        sc[9] = 0.0f; // board games have no rewards (go.h:50)
    }
    waveSync();
    expandBackupBody(pv, nullptr, nullptr, nullptr, nullptr, nullptr, sc + 8 - g, sc + 9 - g, slot, a->err, g, lane, tiles, 2);
}

template <int WPE>
__device__ __forceinline__ void simCandExpandImpl(CSimArgs* __restrict__ a, int rot, int slot, int g, int lane, float* tiles, float* xchg, int part)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
    slot = __builtin_amdgcn_readfirstlane(slot);
    rot = __builtin_amdgcn_readfirstlane(rot);
    const GoDevView gv = simLeafView(ldc(&a->gv), xchg, g);
    const PoolView pv = simPathView(ldc(&a->pv), reinterpret_cast<int*>(xchg) - 2 * a->pv.max_depth - 2, g);
    const SimXchg x{gv.A + (gv.A & 1)};
    const size_t ga = size_t(g) * gv.A;
    float* sc = xchg + x.scalars();
    int* cand_count = reinterpret_cast<int*>(sc + 1) - g;
    int* cand_player = reinterpret_cast<int*>(sc + 2) - g;
    int* cand_action = reinterpret_cast<int*>(xchg + x.caction()) - ga;
    Cand* cs = reinterpret_cast<Cand*>(tiles);
    Cand* out = cs + gv.A;
    const int k = reinterpret_cast<const int*>(sc)[1];
    if (a->cand_coop == 2) {
        if (k > 0 && k <= kCandCoopMaxW) {
            float* dense = simCandDense(tiles, gv.A);
            candScatterW(cs, out, reinterpret_cast<int*>(out + gv.A), k, 8, lane, reinterpret_cast<const int*>(dense + kCandCoopMaxW), a->err);
        }
    } else if (k > 0 && k <= kCandCoopMax && a->cand_coop) {
        float* dense = simCandDense(tiles, gv.A);
        candScatter(cs, out, reinterpret_cast<int*>(out + gv.A), k, 8, lane, reinterpret_cast<const int*>(dense + kCandCoopMax), a->err);
    }
    MZ_LPROF(8);
    azCandStore(gv, sc - g, out, k, cand_count, cand_action, xchg + x.cpolicy() - ga, xchg + x.clogit() - ga, cand_player, sc + 3 - g, sc + 4 - g, g, lane);
    waveSync();
    MZ_LPROF(9);
    expandBackupBody(pv, cand_count, cand_action, xchg + x.cpolicy() - ga, xchg + x.clogit() - ga, cand_player, sc + 3 - g, sc + 4 - g, slot, a->err, g,
                     lane, tiles, part);
    MZ_LPROF(12);
}

template <int WPE>
__device__ __noinline__ void simCandExpand(CSimArgs* __restrict__ a, int rot, int slot, int g, int lane, float* tiles, float* xchg, int part)
{
    simCandExpandImpl<WPE>(a, rot, slot, g, lane, tiles, xchg, part);
}

// Root exploration noise (ref zero_actor.cpp:194-213): policy = (1 - eps) * policy + eps * noise for the root's children, in storage
// order; the noise values were drawn on the host in the reference's RNG order (their count only depends on the number of legal moves)
template <int WPE>
__device__ __noinline__ void simApplyRootNoise(CSimArgs* __restrict__ a, int g, int lane)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
    const PoolView v = ldc(&a->pv);
    const size_t base = size_t(g) * v.cap;
    const int nc = v.rec[base].num_children;
    const size_t fc = base + v.rec[base].first_child;
    const float eps = a->noise_eps;
    for (int i = lane; i < nc; i += 64) {
        const float nz = a->root_noise[size_t(g) * v.A + i];
        if (a->noise_kind == 1) { v.rec[fc + i].policy = (1 - eps) * v.rec[fc + i].policy + eps * nz; }
        else { v.logit[fc + i] = v.logit[fc + i] + nz; }
        v.noise[fc + i] = nz;
    }
    waveSync();
}

// Gumbel: sequential halving + the root child the next simulation starts from (slot >= 1); the first simulation of a launch takes the
// start node the host computed when it ran this step itself (it does at every launch boundary, reading the state back first)
// state_lds: the game's Gumbel state lives in LDS for the launch (sim_kernel_mz) instead of the pool's array
template <int WPE>
__device__ __noinline__ void simGumbelStart(CSimArgs* __restrict__ a, int slot, bool host_start, int g, int lane, float* tiles, int* state_lds = nullptr)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (host_start && slot >= 1) { return; } // a->start[g] was uploaded by the host
    int st = 0;
    if (slot >= 1 && !host_start) {
        const PoolView pv = ldc(&a->pv);
        GumbelView gum = ldc(&a->gum);
        if (state_lds) { gum.state = state_lds - size_t(g) * (3 + kGumbelMaxSample); }
        st = gumbelStepBody(pv, gum, slot, g, lane, tiles);
    }
    if (lane == 0) { a->start[g] = st; }
    waveSync();
}

// the heads read the tower's last activations where they are (an LDS tile); tile 0 (the blocks' temporary) is free for their scratch
template <int WPE, bool BIGA = false, bool FP = BIGA>
__device__ __forceinline__ void simHeadsImpl(CSimArgs* __restrict__ a, int g, int tid, float* tiles, const float* xtile, int xcs, int xpw, float* xchg)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
    xcs = __builtin_amdgcn_readfirstlane(xcs);
    xpw = __builtin_amdgcn_readfirstlane(xpw);
    MZ_HPROF(0);
    const HeadParams hp = ldc(&a->hp);
    const SimXchg x{hp.A + (hp.A & 1)};
    const size_t ga = size_t(g) * hp.A;
    headsBody<BIGA, FP>(nullptr, hp, xchg + x.policy() - ga, xchg + x.logit() - ga, xchg + x.scalars() - g, nullptr, nullptr, 0, g, tid, 512, tiles, xtile, xcs, xpw);
    MZ_HPROF(1);
}
// a non-inlined device function saves the callee-saved VGPRs it uses on entry (scratch stores + loads by all 8 waves): worth it for the
// 9x9 kernels (the heads keep their own register budget, and their version needs none saved), not for the 128-VGPR 8x8 / 3x3 kernels
template <int WPE, bool BIGA = false, bool FP = BIGA>
__device__ __noinline__ void simHeads(CSimArgs* __restrict__ a, int g, int tid, float* tiles, const float* xtile, int xcs, int xpw, float* xchg)
{
    simHeadsImpl<WPE, BIGA, FP>(a, g, tid, tiles, xtile, xcs, xpw, xchg);
}

// 8x8 boards: three tower tiles are 80 KB of LDS, so TWO games share a CU (16 waves) if the kernel stays within 128 VGPRs: one game's
// tree phases and barrier bubbles are filled by the other's tower
// waves per SIMD the kernel is compiled for: 4 (= two resident workgroups per CU, 128 VGPRs) for boards up to 64 points, whose tower
// fits that register budget; 9x9 Go keeps 2 (its 6 pixel tiles per wave pair need ~166 VGPRs, and BASELINE's 256 games are one per CU)
template <int H, int W, int CIN0_PAD, int CPAD, bool BF = false>
constexpr int simWavesPerEu() { return (!BF && H * W <= 64 && kTowerTiles * (CIN0_PAD > CPAD ? CIN0_PAD : CPAD) * planeStride(H, W) * 4 <= 76 * 1024) ? 4 : 2; }
// floats of the LDS region the tower works in (the tree phases and the heads take their scratch from its start)
template <int H, int W, int CIN0_PAD, int CPAD, bool BF>
constexpr int simTileFloats() { return BF ? towerBf16LdsBytes<H, W>(true) / 4 : kTowerTiles * (CIN0_PAD > CPAD ? CIN0_PAD : CPAD) * planeStride(H, W); }

template <int H, int W, int CIN0_PAD, int CPAD>
__device__ __noinline__ const float* simTower(CSimArgs* __restrict__ a, int g, int tid, float* tiles, float* xchg)
{
    // arguments of a device function arrive in VGPRs: tell the compiler which ones are wave-uniform
    g = __builtin_amdgcn_readfirstlane(g);
#ifdef MZ_SIM_TPROF
    if (tid == 0 && g == 0) { s_tp_prev = clock64(); s_tp_idx = 0; g_tp[63] += 1; }
#endif
    g = __builtin_amdgcn_readfirstlane(g);
    const GoDevView gv = simLeafView(ldc(&a->gv), xchg, g);
    return towerBody<H, W, CIN0_PAD, CPAD>(reinterpret_cast<const float*>(gv.feat), a->params, *(const TowerArgs*)&a->ta, nullptr, g, tid, tiles);
}

// the opt-in bf16x3 tower inside the simulation kernel: same hand-over (bit-packed planes in, f32 padded planes out) as simTower
template <int H, int W>
__device__ __noinline__ const float* simTowerBf16(CSimArgs* __restrict__ a, int g, int tid, float* tiles, float* xchg)
{
    g = __builtin_amdgcn_readfirstlane(g);
    const GoDevView gv = simLeafView(ldc(&a->gv), xchg, g);
    return towerBodyBf16<H, W>(reinterpret_cast<const unsigned*>(gv.feat), a->wfrag, a->params, *(const TowerArgsBf16*)&a->tb, nullptr, g, tid,
                               reinterpret_cast<char*>(tiles));
}

template <int H, int W, int CIN0_PAD, int CPAD, int CPL, bool BF = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(simWavesPerEu<H, W, CIN0_PAD, CPAD, BF>(), 4))) void sim_kernel(const SimArgs* __restrict__ a_, const uint8_t* __restrict__ rot_tab, int sim0, int nsims, int host_start)
{
    CSimArgs* a = (CSimArgs*)a_;
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int games = gridDim.x;
    constexpr int WPE = simWavesPerEu<H, W, CIN0_PAD, CPAD, BF>();
    // the reciprocal table of the PUCT divisions lives in LDS above the tower's tiles for the whole launch
    constexpr int kTileFloats = simTileFloats<H, W, CIN0_PAD, CPAD, BF>();
    double* rcp_w = reinterpret_cast<double*>(tiles + kTileFloats);
    for (int i = tid; i < a->rcp_n; i += 512) { rcp_w[i] = a->pv.rcp_tab[i]; }
    __syncthreads();
    LdsCDouble* rcp_lds = (LdsCDouble*)rcp_w;
    // path-speculation memory of the walk (pool_body.h) behind the reciprocal table: LDS copies of the sqrt / bias tables and the remembered
    // paths — only in the one-game-per-CU kernels (9x9: deep principal variations); the 8x8 / 3x3 trees of BASELINE's configs are shallow
    // and those kernels need their LDS to keep two games on a CU
    SpecMem spec{nullptr, nullptr, nullptr};
    int* spec_w = nullptr;
    // the phases' hand-over block (SimXchg) behind whatever the kernel keeps in LDS, with the path of the simulation (2 * max_depth + 2 words) in front
    const int path_words = 2 * a->pv.max_depth + 2;
    float* xchg = reinterpret_cast<float*>(rcp_w + a->rcp_n) + path_words;
    if constexpr (WPE == 2) {
        const int tab_n = a->rcp_n - 2;
        double* sqrt_w = rcp_w + a->rcp_n;
        float* bias_w = reinterpret_cast<float*>(sqrt_w + tab_n);
        spec_w = reinterpret_cast<int*>(bias_w + tab_n + (tab_n & 1));
        for (int i = tid; i < tab_n; i += 512) { sqrt_w[i] = a->pv.sqrt_tab[i]; bias_w[i] = a->pv.bias_tab[i]; }
        if (tid < kSpecWays) { spec_w[tid * kSpecWay] = 0; }
        if (tid < 8) { spec_w[kSpecWays * kSpecWay + tid] = 0; }
        if (tid < kHelpSegs) { spec_w[kSpecHelp + tid * kHelpSeg] = 0; }
        __syncthreads();
        spec = SpecMem{(a->no_spec & 1) ? nullptr : (LdsI32*)spec_w, (LdsCFloat*)bias_w, (LdsCDbl*)sqrt_w};
        xchg = reinterpret_cast<float*>(spec_w + kSpecWords) + path_words;
    }
    // Go, one game per CU: the root's positional-superko table (8 KB, constant during the move) behind the hand-over block
    const uint64_t* seen_lds = nullptr;
    if constexpr (CPL > 0 && WPE == 2) {
        uint64_t* sw = reinterpret_cast<uint64_t*>(xchg + ((simXchgWordsDev(a->gv.A, a->gv.channels, a->gv.W32) + 1) & ~1));
        for (int i = tid; i < kGoSeenCap; i += 512) { sw[i] = a->gv.snap[g].seen[i]; }
        __syncthreads();
        seen_lds = sw;
    }
    // ... and behind it the leaf's scratch block, which then outlives the tower: the part of the leaf only the phases AFTER the network need (path
    // hashes, liberties, legal mask: 4.4 of its 8.1 us on BASELINE configs[1]) runs on waves 6 and 7 beside the heads, in which those waves have no share
    uint64_t* leaf_smem = nullptr;
    if constexpr (CPL > 0 && WPE == 2) {
        const HeadParams hp = ldc(&a->hp);
        if (hp.VH <= 256 && (hp.PC + 1) * hp.P <= 384 && hp.A <= 384 && !(a->no_spec & 8)) { // (MZ_NO_SPEC=8: off)
            leaf_smem = const_cast<uint64_t*>(seen_lds) + kGoSeenCap;
            uint64_t* zk = leaf_smem + goLeafKeyWord(a->gv.Ppad, a->gv.W, a->pv.max_depth); // the block's copy of the Zobrist keys (go_body.h)
            for (int i = tid; i < 2 * a->gv.P; i += 512) { zk[i] = a->gv.key[i]; }
            __syncthreads();
        }
    }
    unsigned long long* prof = a->prof ? a->prof + size_t(g) * 8 : nullptr;
    int* const node_count = reinterpret_cast<int*>(xchg) - 1; // (the spare word of the path block: simPathView)
    if (tid == 0) { *node_count = a->pv.num_nodes[g]; }
    __syncthreads();
    for (int s = 0; s < nsims; ++s) {
        const int slot = sim0 + s; // simulation index within the move = position slot of its leaf
        const int rot = rot_tab[size_t(s) * games + g];
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (prof) { t0 = wall_clock64(); }
        if (wave == 0) {
            if (slot == 1 && a->root_noise) { simApplyRootNoise<WPE>(a, g, lane); }
            if (a->use_gumbel) { simGumbelStart<WPE>(a, slot, s == 0 && host_start != 0, g, lane, tiles); }
            simSelectLeaf<CPL, WPE>(a, rot, slot, g, lane, tiles, rcp_lds, spec, xchg, seen_lds, (a->no_spec & 2) ? 0 : s + 1, leaf_smem);
        } else if (WPE == 2 && wave <= kHelpSegs && spec.w && !(a->no_spec & 2)) { // (MZ_NO_SPEC=2: helper segments off)
            simSelectHelper(a, g, lane, wave, s + 1, rcp_lds, spec);
        }
        __syncthreads();
        if constexpr (CPL > 0) {
            simLeafPlanes<CPL>(a, rot, g, wave, lane, leaf_smem ? leaf_smem : reinterpret_cast<const uint64_t*>(tiles), xchg);
            __syncthreads();
        }
        if (prof) { t1 = wall_clock64(); }
        const float* xt;
        if constexpr (BF) { xt = simTowerBf16<H, W>(a, g, tid, tiles, xchg); }
        else if constexpr (WPE == 4) {
            // The 128-VGPR build (two games per CU): as a function of its own the tower saved and restored 20 callee-saved VGPRs per call — 8 waves x 5 KB each way
            // per simulation, most of the 118 KB of HBM traffic per leaf evaluation that rocprofv3 showed for BASELINE configs[2] (profiles/r04_pmc_c3.json).  Its
            // 2 pixel tiles per wave fit the kernel's own budget.
            const GoDevView gvt = simLeafView(ldc(&a->gv), xchg, g);
            xt = towerBody<H, W, CIN0_PAD, CPAD>(reinterpret_cast<const float*>(gvt.feat), a->params, *(const TowerArgs*)&a->ta, nullptr, g, tid, tiles);
        }
        else { xt = simTower<H, W, CIN0_PAD, CPAD>(a, g, tid, tiles, xchg); } // its own function: its own register budget
        __syncthreads();
        if (prof) { t2 = wall_clock64(); }
        if constexpr (WPE == 4) { simHeadsImpl<WPE>(a, g, tid, tiles, xt, planeStride(H, W), W + 2, xchg); }
        else if (leaf_smem && wave >= 6) { simLeafRest<CPL>(a, rot, slot, g, lane, xchg, seen_lds, leaf_smem, 7 - wave); }
        else { simHeads<WPE, false, (MZ_HEADS_FP != 0)>(a, g, tid, tiles, xt, planeStride(H, W), W + 2, xchg); }
        __syncthreads();
        if (prof) { t3 = wall_clock64(); }
        if (wave == 0) { simCandGather<WPE>(a, rot, g, lane, tiles, xchg); }
        __syncthreads();
        {
            const int A = a->gv.A;
            const SimXchg x{A + (A & 1)};
            if (a->cand_coop) { simCandRank(A, reinterpret_cast<const int*>(xchg + x.scalars())[1], wave, lane, tiles); }
        }
        __syncthreads();
        {
            const bool split = !a->pv.value_rescale; // backup beside expand on a second wave
            if (wave == 0) { simCandExpand<WPE>(a, rot, slot, g, lane, tiles, xchg, split ? 1 : 0); }
            else if (wave == 1 && split) { simBackupOnly<WPE>(a, slot, g, lane, tiles, xchg); }
        }
        __syncthreads();
        if (prof && tid == 0) {
            t4 = wall_clock64();
            prof[0] += t1 - t0; prof[1] += t2 - t1; prof[2] += t3 - t2; prof[3] += t4 - t3; prof[4] += 1;
        }
    }
    if (tid == 0) { a->pv.num_nodes[g] = *node_count; }
    if (prof && tid == 0 && spec_w) {
        prof[7] += (static_cast<unsigned long long>(spec_w[kSpecWays * kSpecWay + 1]) << 40) | (static_cast<unsigned long long>(spec_w[kSpecWays * kSpecWay + 5]) << 20) | spec_w[kSpecWays * kSpecWay + 3];
        prof[6] += static_cast<unsigned long long>(spec_w[kSpecWays * kSpecWay + 7]) << 40; // levels taken over from the helper waves (the low bits hold the path lengths)
    }
}


} // namespace mz
