// Search pool of libmzgpu: `games` MCTS trees as one structure-of-arrays node pool in HBM.
// Replaces Tree / MCTSNode / MCTS of the reference (ref actor/tree.h:32-122, actor/mcts.h:17-119,
// actor/mcts.cpp:20-228).  One wave64 per game; see pool.hip for the kernels.
#pragma once
#include "common.h"

namespace mz {

// count/mean/policy/reward: the f32 statistics of ref mcts.h:54-63; action = action_id (root: -1);
// players = own action player | (player of this node's children << 8)
struct __attribute__((aligned(16))) NodeRec {
    float count, mean, policy, reward;
    int first_child, num_children, action, players;
};

struct PoolView { // device pointers, passed to kernels by value
    int games, cap, A, max_depth;
    // per node [games*cap]: a 32-byte "hot" record with everything the PUCT walk needs — the children scan of a level also
    // brings in each child's own (first_child, num_children), so a level costs ONE dependent load — plus cold arrays
    NodeRec* rec;
    float *logit, *noise, *value;
    int* hslot; // MuZero hidden-state slot
    // per game
    int *num_nodes, *path_len, *path, *path_action; // path[g*max_depth + d]
    int *host_path_len, *host_path_action;           // optional host-mapped (pinned) mirrors written by select_kernel (zero-copy D2H)
    float *bound_key;                                // [games][bound_cap] value-bound multiset keys (ref mcts.cpp:219-228)
    int *bound_cnt, *bound_size;                     // [games][bound_cap], [games]
    float *bound_lo, *bound_hi;                      // [games]
    int bound_cap;
    // PUCT tables indexed by N = total_simulation (host-computed with the host libm: bit-exact by construction)
    const float* bias_tab;
    const double* sqrt_tab;
    const double* rcp_tab; // [n + 4] correctly rounded 1/i (index 0 unused): the divisions by visit counts in select_kernel
    float gamma;
    int value_rescale, flipping_player, atari_init_q;
};

class Pool {
public:
    ~Pool();
    int init(int device, int games, int nodes_per_game, int action_size, const mz_search_cfg& cfg, hipStream_t shared_stream);
    int resetSearch(const uint8_t* mask, const int* root_player);
    int select(const int* start_node, int* path_len, int* paths, int* path_action);
    int expandBackup(const int* cand_count, const int* cand_action, const float* cand_policy, const float* cand_logit, const int* cand_player,
                     const float* value, const float* reward);
    int rootSetNoise(const uint8_t* mask, const float* policy, const float* logit, const float* noise);
    int rootRead(int* num_children, int* action, float* count, float* mean, float* policy, float* logit, float* noise, float* value, float* reward,
                 float* root_count, float* root_mean, float* root_value, float* bound_lo, float* bound_hi, int* bound_size, bool launched_ahead = false);
    int rootReadLaunch(); // rootRead's kernel (+ copies) queued without waiting; the next rootRead(..., launched_ahead = true) only waits and unpacks
    int readNodes(int game, int n, int* action, int* player, int* num_children, int* first_child, float* mean, float* count, float* policy,
                  float* logit, float* noise, float* value, float* reward);
    int numNodes(int game);

    // async building blocks for the worker (device-resident arguments, no host sync)
    int selectAsync(const int* d_start_node);
    int expandBackupAsync(int hslot, bool from_host = false); // consumes d_cand_* staging below (or the pinned h_cand_* views directly)
    bool zero_copy_ = false;                        // kernels read/write the pinned host staging directly over PCIe: no memcpy ops per cycle
    int expandBackupStaged(int hslot);             // H2D of the pinned h_cand_* mirrors + expandBackupAsync
    int hiddenIndexAsync(int slots_per_game, int dst_slot, int* d_src_idx, int* d_dst_idx, int* d_action_ids); // MuZero: slots of the last select
    int checkError();                              // device-side error flag (capacity)
    int* errFlag() { return game_i_.p + size_t(v_.games) * 3; }
    int rcpEntries() const { return static_cast<int>(rcp_tab_.n); }
    // low-latency completion: a 1-thread kernel stores `value` into a pinned host word behind everything already queued on
    // the stream; the host spins on that word instead of going through hipStreamSynchronize (falls back to it after 20 ms)
    int signalAsync(int value);
    int waitSignal(int value);
    PinBuf<int> h_flag_;
    int hslot_next_ = -1;

    PoolView v_{};
    int device_ = -1;
    hipStream_t stream_ = nullptr;
    bool own_stream_ = false;
    mz_search_cfg cfg_{};

    // pinned host mirrors + device staging of the per-cycle candidate lists.  The seven candidate arrays are views
    // into ONE pinned arena / ONE device arena (count, player, value, reward, action, policy, logit) so a cycle needs a
    // single H2D copy; path_len + path_action (+ path) likewise share one device arena and one D2H copy.
    template <class T>
    struct View { T* p = nullptr; size_t n = 0; };
    PinBuf<uint32_t> h_cand_arena_, h_path_arena_;
    DevBuf<uint32_t> d_cand_arena_, d_path_arena_;
    View<int> h_cand_count_, h_cand_action_, h_cand_player_, h_path_len_, h_path_, h_path_action_;
    View<float> h_cand_policy_, h_cand_logit_, h_value_, h_reward_;
    View<int> d_cand_count_, d_cand_action_, d_cand_player_;
    View<float> d_cand_policy_, d_cand_logit_, d_value_, d_reward_;
    PinBuf<int> h_start_;
    DevBuf<int> d_start_, d_mask_;
    // root read staging
    DevBuf<float> d_rr_f_; // 8 arrays [games*A] + 5 arrays [games]
    DevBuf<int> d_rr_i_;   // num_children, action[games*A], bound_size
    PinBuf<float> h_rr_f_;
    PinBuf<int> h_rr_i_;

private:
    DevBuf<NodeRec> rec_;
    DevBuf<float> f_nodes_;  // 3 cold float arrays
    DevBuf<int> i_nodes_;    // hslot
    DevBuf<int> game_i_;     // num_nodes, (unused), bound_size, err
    DevBuf<int> bound_cnt_;
    DevBuf<float> bound_key_, game_f_;
    DevBuf<float> bias_tab_;
    DevBuf<double> sqrt_tab_, rcp_tab_;
};

} // namespace mz
