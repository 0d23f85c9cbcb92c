// Device-resident leaf environment for AlphaZero Go: the rules engine, feature planes, legal mask and the
// candidate list (legal filter + inverse rotation + the reference's std::sort order) run on the GPU, so a whole move
// (n + 1 lock-step cycles) is enqueued without a host hop.  Replaces, for env_game=go, the host replay of
// ZeroActor::getEnvironmentTransition / calculateAlphaZeroActionPolicy (ref actor/zero_actor.cpp:55,79,215-252) and
// GoEnv::act / isLegalAction / getFeatures / isTerminal / getEvalScore (ref environment/go/go.cpp:132-308,703-723).
//
// Instead of replaying root -> leaf on a copy of the root environment (O(depth) per simulation), every expanded node keeps
// its POSITION in a slab ([games][n+1] slots, like the MuZero hidden-state slab): stones, Zobrist hash, group id per
// point.  A leaf is its parent's slot + one move: O(1) wave-parallel passes (merge labels, liberty test of the <= 4
// adjacent enemy groups, capture), the history planes come from the slots of the path nodes (older ones from the root's
// 8-ring), the positional-superko set is the root's hash table plus the hashes along the path.
#pragma once
#include "common.h"
#include "pool.h"

namespace mz {

constexpr int kGoMaxN = 19, kGoMaxP = kGoMaxN * kGoMaxN, kGoMaxW = (kGoMaxP + 63) / 64, kGoSeenCap = 1024;
constexpr int kRotPackGames = 1040; // 3 bits per game, 10 games per word

// host -> device once per move per game; filled by the host engine (env.cpp Go::exportDeviceRoot), which stays authoritative
struct GoRootSnapshot {
    uint64_t stones[2][kGoMaxW];    // bit p of word p >> 6 (black, white)
    uint64_t hist[8][2][kGoMaxW];   // ring of the last 8 positions: entry (hist_len - 1 - j) & 7 is j moves ago
    uint64_t seen[kGoSeenCap];      // positional-superko set: open addressing, linear probing, 0 = empty, hash 0 stored as 1
    uint64_t hash;
    int32_t hist_len, turn, nmoves, passes; // passes = trailing consecutive passes (capped at 2)
    uint16_t lab[kGoMaxP + 3];      // group id per point (valid where a stone is): any point of the group
};

struct RotPack { uint32_t w[kRotPackGames / 10]; }; // per-game feature rotation of one cycle, passed as a kernel argument
inline void rotPackSet(RotPack& r, int g, int rot) { r.w[g / 10] = (r.w[g / 10] & ~(7u << (3 * (g % 10)))) | (uint32_t(rot) << (3 * (g % 10))); }

struct GoDevView {
    int kind;                  // 0: Go, 1: Othello (two bitboards + pass count per slot, no hash / group ids; same outputs), 2: TicTacToe
    int channels;              // feature planes of the game (Go 18, Othello 4)
    int games, n, P, W, A, slots, Ppad, W32, LW;
    float komi;
    uint64_t* stones;          // [games][slots][2][W]
    uint64_t* hash;            // [games][slots]
    int* meta;                 // [games][slots][2]: moves played, trailing passes
    uint16_t* lab;             // [games][slots][Ppad]
    const GoRootSnapshot* snap; // [games]
    const uint64_t* key;       // [2][P]
    uint64_t turn_key;         // situational superko (env_go_ko_rule): XORed into the hash on every move, pass included; 0 = positional
    const uint16_t* inv;       // [8][P]  feature rotation: plane bit p <- position inv[r][p]
    const uint16_t* fwd;       // [8][A]  policy index of action a under rotation r
    uint32_t* feat;            // [games][18 * W32] bit-packed planes (the tower's input format)
    uint64_t* legal;           // [games][LW] bit a = action a legal for the player to move at the leaf
    int* leaf_player;          // [games]
    int* terminal;             // [games]
    float* eval;               // [games]
};

class GoDevice {
public:
    // kind 0: Go (keys = Zobrist table [2][P]); kind 1: Othello (board_n <= 8, keys unused); kind 2: TicTacToe (3x3, 9 actions)
    int init(int device, int games, int board_n, float komi, int action_size, int slots, int max_depth, hipStream_t stream, const int* const inv[8],
             const int* const fwd[8], const uint64_t* keys, int kind = 0, uint64_t turn_key = 0);
    GoRootSnapshot* hostSnap(int g) { return h_snap_.p + g; }
    int uploadRoots();                                                   // snapshots H2D + slot 0 of every game
    int leafAsync(const PoolView& pv, const RotPack& rot, int slot);      // position + planes + legal mask of the selected leaves
    // candidate lists of the leaves from the heads' outputs, into the pool's device staging (consumed by expandBackupAsync)
    int candAsync(Pool& pool, const float* d_policy, const float* d_logit, const float* d_value, const RotPack& rot);
    // test access (synchronous): outputs of the last leafAsync
    int readLeaf(uint32_t* feat, uint8_t* legal, int* terminal, float* eval, int* player);
    GoDevView v_{};
    int max_depth_ = 0;
    hipStream_t stream_ = nullptr;

private:
    int device_ = 0;
    PinBuf<GoRootSnapshot> h_snap_;
    DevBuf<GoRootSnapshot> d_snap_;
    DevBuf<uint64_t> stones_, hash_, key_, legal_;
    DevBuf<int> meta_, misc_i_;
    DevBuf<uint16_t> lab_, inv_, fwd_;
    DevBuf<uint32_t> feat_;
    DevBuf<float> eval_;
};

// stand-alone ordering of one candidate list by the device path (tests): out_order[i] = index of the i-th candidate
int sortCandidatesOnDevice(int device, const float* policy, int n, int* out_order);

} // namespace mz
