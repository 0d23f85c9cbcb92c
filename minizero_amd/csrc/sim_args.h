// The argument block of the per-game simulation kernels (sim.hip) and of the batched evaluation of Gumbel rounds (sim_rounds.hip): one structure in device
// memory, read through the constant address space.
#pragma once
#include "net.h"
#include "net_body.h"
#include "net_bf16_body.h"
#include "net_atari_body.h"
#include "pool_body.h"
#include "go_body.h"
#include "gumbel_body.h"
#include <type_traits>

namespace mz {

struct SimArgs {
    PoolView pv;
    GoDevView gv;
    TowerArgs ta;
    HeadParams hp;
    const float* params;
    float* act;                       // [games][C][P] tower output (input of the heads); the one-tile tower's block x (net_wide_body.h)
    float* act2;                      // [games][C][P] the one-tile tower's temporary
    float *policy, *logit, *value;    // heads outputs
    int *cand_count, *cand_action, *cand_player;
    float *cand_policy, *cand_logit, *value_io, *reward_io;
    int* err;
    int rcp_n;                        // entries of pv.rcp_tab
    // MuZero (sim_kernel_mz): dynamics trunk, hidden-state slab, root planes / legal mask / player from the host engine
    TowerArgs ta_dyn;
    float* hidden;                    // [games][slots][C * P]
    const unsigned* root_feat;        // [games][cin * ceil(P / 32)] bit-packed planes of the root position
    const unsigned long long* root_legal; // [games][LW] legal mask of the root position
    const int* root_turn;             // [games] player to move at the root
    int slots, A, LW, num_players;
    const float* root_noise;          // [games][A] noise of the root children (host RNG), applied before simulation 1; nullptr: none
    float noise_eps;
    int noise_kind;                   // 1: Dirichlet on the priors, 2: Gumbel on the logits (ref zero_actor.cpp:194-213)
    int use_gumbel;                   // Gumbel root logic (sequential halving + start node) between simulations
    GumbelView gum;
    int* start;                       // [games] start node of the next selection (written by the Gumbel step or by the host)
    // muzero_atari (sim_kernel_mz, simulations >= 1; the 96x96 representation of the root runs as stand-alone kernels)
    int atari, action_planes;
    AtariHeadParams ahp;
    float* reward;                    // [games] reward head output (game scale)
    int no_spec;                      // MZ_NO_SPEC=1: path speculation of the walk off (experiments)
    int cand_coop;                    // the candidate rank sort is shared by the 8 waves (its scratch fits the tower tiles)
    // opt-in bf16x3 tower (net_bf16_body.h): fragments + layer table; used by the BF instantiations of sim_kernel
    const uint4* wfrag;
    TowerArgsBf16 tb;
    unsigned* cluster;                // cluster mode (sim_cluster.h): per-game exchange block of `cluster_words` words; nullptr: one workgroup per game
    int cluster_words, oct_words;
    unsigned* cluster_oct;            // cluster mode: the blocks of the octet-wide 601-bin heads ([8 octets][2 heads][oct_words]); nullptr: per-game heads
    unsigned long long* prof;         // optional (MZ_SIM_PROF=1): per game, 100-MHz ticks spent in [select+leaf, tower, heads, cand+expand] + sims
    // leaves evaluated AHEAD of their simulations (sim_pre_kernel_mz below): one entry per (game, slot of the simulation) of the current move
    int* pre_key;                     // [games][slots][4] = {parent's slab slot, action, epoch of the move, -}
    float *pre_policy, *pre_logit;    // [games][slots][A]: the leaf's children in the reference's sort order (policy descending, zero_actor.cpp:241-243) ...
    int* pre_action;                  // ... and their actions: the candidate list is built where the leaf was evaluated, not in the in-order part
    float *pre_value, *pre_reward;    // [games][slots], game scale
    unsigned* pre_stat;               // [0] simulations that found their leaf evaluated, [1] leaves evaluated ahead (tests / monitoring)
    int alt_base;                     // != 0: slots alt_base + s hold a SECOND expected leaf of simulation s (sim_pre_kernel_mz, hypothesis 1)
};

// SimArgs never changes during a launch: the device functions read it through the CONSTANT address space, i.e. with scalar loads whose
// results are wave-uniform and can be kept / re-used across stores.  Through a generic pointer every field was a flat load (divergent for
// the compiler, since a flat address may be private memory: the whole selection loop was compiled with exec-mask control flow) that had
// to be repeated after every store — the PUCT walk waited for such a reload on every level.
typedef __attribute__((address_space(4))) const SimArgs CSimArgs;
template <class T>
__device__ __forceinline__ T ldc(__attribute__((address_space(4))) const T* p) // by-value copy of a sub-structure (SGPRs after SROA)
{
    static_assert(sizeof(T) % 4 == 0 && std::is_trivially_copyable<T>::value, "word-copied");
    typedef __attribute__((address_space(4))) const unsigned CU;
    CU* s = (CU*)p;
    unsigned w[sizeof(T) / 4];
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) { w[i] = s[i]; }
    T t;
    __builtin_memcpy(&t, w, sizeof(T));
    return t;
}

// ... and so does the path of the simulation (node ids, moves, length): written by the walk, read by the leaf and by expand + backup; the word behind
// them holds the game's node count for the launch (expand reads and advances it at every simulation: simNodeCountIn / simNodeCountOut)
__device__ __forceinline__ PoolView simPathView(PoolView pv, int* lds_path, int g)
{
    const size_t off = size_t(g) * pv.max_depth;
    pv.path = lds_path - off;
    pv.path_action = lds_path + pv.max_depth - off;
    pv.path_len = lds_path + 2 * pv.max_depth - g;
    pv.num_nodes = lds_path + 2 * pv.max_depth + 1 - g;
    pv.host_path_len = nullptr;
    pv.host_path_action = nullptr;
    return pv;
}

// The same view for use INSIDE a kernel body (or a function inlined into one), where the compiler can see that `lds_path` is LDS: the biased pointers above lie
// in front of the block (undefined behaviour by the letter), and with the provenance in sight the compiler may fold the bias into a DS instruction's offset — the
// address then leaves the LDS as soon as g * max_depth * 4 exceeds the block's LDS offset (260 Atari-shaped games at n = 50; never with BASELINE's 64): reads return
// nothing, the slab index built from them faults.  Laundering the pointers makes them opaque generic pointers: 64-bit arithmetic, flat accesses, always in the aperture.
template <class T>
__device__ __forceinline__ T* opaquePtr(T* p)
{
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ PoolView simPathViewSafe(PoolView pv, int* lds_path, int g)
{
    pv = simPathView(pv, opaquePtr(lds_path), g);
    return pv;
}

} // namespace mz
