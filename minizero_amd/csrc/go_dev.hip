// Device-resident Go leaf environment + candidate lists (see go_dev.h).  One wave64 per game; point p = i * 64 + lane is
// cell i of lane `lane`, so a ballot over cell i is word i of a bitboard.  All group bookkeeping is O(1) passes over the
// cells: no flood fill, no replay.  Integer / bit work only (bit-exact against the host engine by construction of the
// same rules; checked by tests/test_gpu_godev.py playouts and end to end by tests/test_gpu_worker.py).
#include "go_body.h"
#include <cstring>
#include <vector>

namespace mz {

namespace {

// snapshot -> slot 0 of the game's position slab
__global__ __launch_bounds__(64) void go_root_kernel(GoDevView v)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const GoRootSnapshot& S = v.snap[g];
    const size_t sb = size_t(g) * v.slots;
    if (lane < 2 * v.W) { v.stones[sb * 2 * v.W + lane] = S.stones[lane / v.W][lane % v.W]; }
    if (lane == 0) {
        v.hash[sb] = S.hash;
        v.meta[sb * 2] = S.nmoves;
        v.meta[sb * 2 + 1] = S.passes;
    }
    for (int p = lane; p < v.P; p += 64) { v.lab[sb * v.Ppad + p] = S.lab[p]; }
}

template <int CPL>
__global__ __launch_bounds__(64) void go_leaf_kernel(GoDevView v, PoolView pv, RotPack rp, int slot)
{
    extern __shared__ uint64_t smem[];
    goLeafBody<CPL>(v, pv, rotOf(rp, blockIdx.x), slot, blockIdx.x, threadIdx.x, smem);
}

__global__ __launch_bounds__(64) void ttt_leaf_kernel(GoDevView v, PoolView pv, RotPack rp, int slot)
{
    tttLeafBody(v, pv, rotOf(rp, blockIdx.x), slot, blockIdx.x, threadIdx.x);
}

__global__ __launch_bounds__(64) void oth_leaf_kernel(GoDevView v, PoolView pv, RotPack rp, int slot)
{
    othLeafBody(v, pv, rotOf(rp, blockIdx.x), slot, blockIdx.x, threadIdx.x);
}

__global__ __launch_bounds__(64) void az_cand_kernel(GoDevView v, const float* __restrict__ policy, const float* __restrict__ logit,
                                                     const float* __restrict__ value, RotPack rp, int* __restrict__ cand_count,
                                                     int* __restrict__ cand_action, float* __restrict__ cand_policy, float* __restrict__ cand_logit,
                                                     int* __restrict__ cand_player, float* __restrict__ value_out, float* __restrict__ reward_out,
                                                     int* __restrict__ err)
{
    extern __shared__ uint64_t smem[];
    azCandBody(v, policy, logit, value, rotOf(rp, blockIdx.x), cand_count, cand_action, cand_policy, cand_logit, cand_player, value_out, reward_out, err,
               blockIdx.x, threadIdx.x, smem);
}

__global__ __launch_bounds__(64) void sort_test_kernel(const float* __restrict__ policy, int n, int* __restrict__ order, int* __restrict__ err)
{
    extern __shared__ uint64_t smem[];
    Cand* cs = reinterpret_cast<Cand*>(smem);
    Cand* out = cs + n;
    int* stack = reinterpret_cast<int*>(out + n);
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) { cs[i] = Cand{i, policy[i], 0.0f}; }
    waveSync();
    orderCandidates(cs, out, stack, n, lane, err);
    for (int i = lane; i < n; i += 64) { order[i] = out[i].action; }
}

} // namespace

// ------------------------------------------------------------------------------------------------
int GoDevice::init(int device, int games, int board_n, float komi, int action_size, int slots, int max_depth, hipStream_t stream, const int* const inv[8],
                   const int* const fwd[8], const uint64_t* keys, int kind, uint64_t turn_key)
{
    if (kind == 1 && board_n > 8) { setError("GoDevice: Othello boards up to 8x8"); return MZ_ERR_ARG; }
    if (kind == 2 && (board_n != 3 || action_size != 9)) { setError("GoDevice: TicTacToe is 3x3 with 9 actions"); return MZ_ERR_ARG; }
    if (board_n < 2 || board_n > kGoMaxN || games < 1 || games > kRotPackGames || action_size != board_n * board_n + (kind == 2 ? 0 : 1)) {
        setError("GoDevice: unsupported shape (board %d, %d games, %d actions)", board_n, games, action_size);
        return MZ_ERR_ARG;
    }
    device_ = device;
    stream_ = stream;
    max_depth_ = max_depth;
    MZ_HIP(hipSetDevice(device));
    GoDevView& v = v_;
    v.kind = kind; v.channels = kind == 0 ? 18 : 4;
    v.games = games; v.n = board_n; v.P = board_n * board_n; v.W = (v.P + 63) / 64; v.A = action_size; v.slots = slots;
    v.Ppad = 64 * v.W; v.W32 = (v.P + 31) / 32; v.LW = (v.A + 63) / 64; v.komi = komi;
    const size_t GS = size_t(games) * slots;
    if (!h_snap_.alloc(games) || !d_snap_.alloc(games) || !stones_.alloc(GS * 2 * v.W) || !hash_.alloc(GS) || !meta_.alloc(GS * 2) ||
        !lab_.alloc(GS * v.Ppad) || !key_.alloc(size_t(2) * v.P) || !inv_.alloc(size_t(8) * v.P) || !fwd_.alloc(size_t(8) * v.A) ||
        !feat_.alloc(size_t(games) * v.channels * v.W32) || !legal_.alloc(size_t(games) * v.LW) || !misc_i_.alloc(size_t(games) * 2) || !eval_.alloc(games)) {
        setError("GoDevice: allocation failed");
        return MZ_ERR_DEVICE;
    }
    memset(h_snap_.p, 0, games * sizeof(GoRootSnapshot));
    MZ_HIP(hipMemset(stones_.p, 0, stones_.n * sizeof(uint64_t)));
    MZ_HIP(hipMemset(lab_.p, 0, lab_.n * sizeof(uint16_t)));
    std::vector<uint16_t> t(size_t(8) * v.A);
    for (int r = 0; r < 8; ++r) { for (int p = 0; p < v.P; ++p) { t[size_t(r) * v.P + p] = static_cast<uint16_t>(inv[r][p]); } }
    MZ_HIP(hipMemcpy(inv_.p, t.data(), size_t(8) * v.P * sizeof(uint16_t), hipMemcpyHostToDevice));
    for (int r = 0; r < 8; ++r) { for (int a = 0; a < v.A; ++a) { t[size_t(r) * v.A + a] = static_cast<uint16_t>(fwd[r][a]); } }
    MZ_HIP(hipMemcpy(fwd_.p, t.data(), size_t(8) * v.A * sizeof(uint16_t), hipMemcpyHostToDevice));
    if (keys) { MZ_HIP(hipMemcpy(key_.p, keys, size_t(2) * v.P * sizeof(uint64_t), hipMemcpyHostToDevice)); }
    v.stones = stones_.p; v.hash = hash_.p; v.meta = meta_.p; v.lab = lab_.p; v.snap = d_snap_.p; v.key = key_.p; v.inv = inv_.p; v.fwd = fwd_.p;
    v.turn_key = turn_key;
    v.feat = feat_.p; v.legal = legal_.p; v.leaf_player = misc_i_.p; v.terminal = misc_i_.p + games; v.eval = eval_.p;
    return MZ_OK;
}

int GoDevice::uploadRoots()
{
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipMemcpyAsync(d_snap_.p, h_snap_.p, v_.games * sizeof(GoRootSnapshot), hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(go_root_kernel, dim3(v_.games), dim3(64), 0, stream_, v_);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int GoDevice::leafAsync(const PoolView& pv, const RotPack& rot, int slot)
{
    if (slot < 0 || slot >= v_.slots) { setError("GoDevice::leafAsync: slot %d out of range", slot); return MZ_ERR_ARG; }
    if (v_.kind == 2) {
        hipLaunchKernelGGL(ttt_leaf_kernel, dim3(v_.games), dim3(64), 0, stream_, v_, pv, rot, slot);
        MZ_HIP(hipGetLastError());
        return MZ_OK;
    }
    if (v_.kind == 1) {
        hipLaunchKernelGGL(oth_leaf_kernel, dim3(v_.games), dim3(64), 0, stream_, v_, pv, rot, slot);
        MZ_HIP(hipGetLastError());
        return MZ_OK;
    }
    const size_t smem = goLeafSmemBytes(v_, pv.max_depth);
#define MZ_GO_CASE(K) \
    case K: hipLaunchKernelGGL(go_leaf_kernel<K>, dim3(v_.games), dim3(64), smem, stream_, v_, pv, rot, slot); break;
    switch (v_.W) {
        MZ_GO_CASE(1) MZ_GO_CASE(2) MZ_GO_CASE(3) MZ_GO_CASE(4) MZ_GO_CASE(5) MZ_GO_CASE(6)
    default: setError("GoDevice: board too large"); return MZ_ERR_ARG;
    }
#undef MZ_GO_CASE
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int GoDevice::candAsync(Pool& pool, const float* d_policy, const float* d_logit, const float* d_value, const RotPack& rot)
{
    const size_t smem = azCandSmemBytes(v_.A);
    hipLaunchKernelGGL(az_cand_kernel, dim3(v_.games), dim3(64), smem, stream_, v_, d_policy, d_logit, d_value, rot, pool.d_cand_count_.p,
                       pool.d_cand_action_.p, pool.d_cand_policy_.p, pool.d_cand_logit_.p, pool.d_cand_player_.p, pool.d_value_.p, pool.d_reward_.p,
                       pool.errFlag());
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int GoDevice::readLeaf(uint32_t* feat, uint8_t* legal, int* terminal, float* eval, int* player)
{
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipStreamSynchronize(stream_));
    const int G = v_.games;
    if (feat) { MZ_HIP(hipMemcpy(feat, v_.feat, size_t(G) * v_.channels * v_.W32 * sizeof(uint32_t), hipMemcpyDeviceToHost)); }
    if (legal) {
        std::vector<uint64_t> w(size_t(G) * v_.LW);
        MZ_HIP(hipMemcpy(w.data(), v_.legal, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (int g = 0; g < G; ++g) {
            for (int a = 0; a < v_.A; ++a) { legal[size_t(g) * v_.A + a] = (w[size_t(g) * v_.LW + (a >> 6)] >> (a & 63)) & 1; }
        }
    }
    if (terminal) { MZ_HIP(hipMemcpy(terminal, v_.terminal, G * sizeof(int), hipMemcpyDeviceToHost)); }
    if (player) { MZ_HIP(hipMemcpy(player, v_.leaf_player, G * sizeof(int), hipMemcpyDeviceToHost)); }
    if (eval) { MZ_HIP(hipMemcpy(eval, v_.eval, G * sizeof(float), hipMemcpyDeviceToHost)); }
    return MZ_OK;
}

int sortCandidatesOnDevice(int device, const float* policy, int n, int* out_order)
{
    if (n < 1 || n > 4096) { setError("sortCandidatesOnDevice: n out of range"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device));
    DevBuf<float> dp;
    DevBuf<int> dord;
    if (!dp.alloc(n) || !dord.alloc(n + 1)) { setError("sortCandidatesOnDevice: allocation failed"); return MZ_ERR_DEVICE; }
    MZ_HIP(hipMemcpy(dp.p, policy, n * sizeof(float), hipMemcpyHostToDevice));
    MZ_HIP(hipMemset(dord.p + n, 0, sizeof(int)));
    hipLaunchKernelGGL(sort_test_kernel, dim3(1), dim3(64), 2 * size_t(n) * sizeof(Cand) + kSortStackBytes + 16, nullptr, dp.p, n, dord.p, dord.p + n);
    MZ_HIP(hipGetLastError());
    MZ_HIP(hipDeviceSynchronize());
    int err = 0;
    MZ_HIP(hipMemcpy(out_order, dord.p, n * sizeof(int), hipMemcpyDeviceToHost));
    MZ_HIP(hipMemcpy(&err, dord.p + n, sizeof(int), hipMemcpyDeviceToHost));
    if (err) { setError("sortCandidatesOnDevice: device sort failed"); return err; }
    return MZ_OK;
}

} // namespace mz
