// gfx950 kernels of the search pool.  One wave64 per game (grid = games, block = 64):
//   select_kernel         wavefront-parallel PUCT arg-max walk root -> leaf (ref actor/mcts.cpp:139-149,181-217)
//   expand_backup_kernel  bump-allocate + write children (coalesced SoA), then the leaf -> root running-mean
//                         chain and the value-bound multiset (ref mcts.cpp:151-179,219-228)
// Bit-exactness: built with -ffp-contract=off; the log()/sqrt() of the PUCT bias come from host-computed
// tables indexed by N (ref mcts.cpp:57-58 uses the double libm functions); the init-Q sum is an ordered
// f32 sum over the visited children in storage order; ties follow mcts.cpp:191.
#include "pool_body.h"
#include <cfloat>
#include <climits>
#include <cmath>
#include <atomic>
#include <cstring>

namespace mz {

__global__ void reset_kernel(PoolView v, const int* __restrict__ mask, const int* __restrict__ root_player)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= v.games || (mask && !mask[g])) { return; }
    const size_t r = size_t(g) * v.cap;
    NodeRec n;
    n.count = 0; n.mean = 0; n.policy = 0; n.reward = 0;
    n.first_child = -1; n.num_children = 0; n.action = -1;
    n.players = root_player[g];
    v.rec[r] = n;
    v.logit[r] = 0; v.noise[r] = 0; v.value[r] = 0; v.hslot[r] = -1;
    v.num_nodes[g] = 1;
    v.path_len[g] = 0;
    v.bound_size[g] = 0;
    v.bound_lo[g] = 0; v.bound_hi[g] = 0;
}

__global__ __launch_bounds__(64) void select_kernel(PoolView v, const int* __restrict__ start) { selectBody(v, start, blockIdx.x, threadIdx.x, v.rcp_tab); }

__global__ __launch_bounds__(64) void expand_backup_kernel(PoolView v, const int* __restrict__ cand_count, const int* __restrict__ cand_action,
                                                           const float* __restrict__ cand_policy, const float* __restrict__ cand_logit,
                                                           const int* __restrict__ cand_player, const float* __restrict__ value_in,
                                                           const float* __restrict__ reward_in, int hslot, int* __restrict__ err)
{
    extern __shared__ float lds[]; // value-bound multiset: keys [bound_cap] then counts [bound_cap] (only with value_rescale)
    expandBackupBody(v, cand_count, cand_action, cand_policy, cand_logit, cand_player, value_in, reward_in, hslot, err, blockIdx.x, threadIdx.x, lds);
}

__global__ __launch_bounds__(64) void root_set_noise_kernel(PoolView v, const int* __restrict__ mask, const float* __restrict__ policy,
                                                            const float* __restrict__ logit, const float* __restrict__ noise)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    if (mask && !mask[g]) { return; }
    const size_t base = size_t(g) * v.cap;
    const int nc = v.rec[base].num_children;
    const size_t fc = base + v.rec[base].first_child;
    for (int i = lane; i < nc; i += 64) {
        const size_t c = size_t(g) * v.A + i;
        v.rec[fc + i].policy = policy[c];
        v.logit[fc + i] = logit[c];
        v.noise[fc + i] = noise[c];
    }
}

// gather the root's children into compact [games][A] arrays (f: action-major block of 7 float arrays, then 5 per-game arrays)
__global__ __launch_bounds__(64) void root_read_kernel(PoolView v, float* __restrict__ f, int* __restrict__ iv)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const size_t base = size_t(g) * v.cap, GA = size_t(v.games) * v.A;
    const NodeRec root = v.rec[base];
    const int nc = root.num_children;
    const size_t fc = base + (nc > 0 ? root.first_child : 0);
    for (int i = lane; i < nc; i += 64) {
        const size_t c = size_t(g) * v.A + i;
        const NodeRec r = v.rec[fc + i];
        f[0 * GA + c] = r.count;
        f[1 * GA + c] = r.mean;
        f[2 * GA + c] = r.policy;
        f[3 * GA + c] = v.logit[fc + i];
        f[4 * GA + c] = v.noise[fc + i];
        f[5 * GA + c] = v.value[fc + i];
        f[6 * GA + c] = r.reward;
        iv[v.games + c] = r.action;
    }
    if (lane == 0) {
        float* pg = f + 7 * GA;
        pg[0 * v.games + g] = root.count;
        pg[1 * v.games + g] = root.mean;
        pg[2 * v.games + g] = v.value[base];
        pg[3 * v.games + g] = v.bound_lo[g];
        pg[4 * v.games + g] = v.bound_hi[g];
        iv[g] = nc;
        iv[v.games + GA + g] = v.bound_size[g];
    }
}

__global__ void signal_kernel(int* flag, int value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// MuZero: hidden-state slab slots for the leaves of the last select (parent slot -> source, dst_slot -> destination)
__global__ void hidden_index_kernel(PoolView v, int slots_per_game, int dst_slot, int* __restrict__ src_idx, int* __restrict__ dst_idx,
                                    int* __restrict__ action_ids)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= v.games) { return; }
    const size_t base = size_t(g) * v.cap;
    const int len = v.path_len[g];
    const int* path = v.path + size_t(g) * v.max_depth;
    const int leaf = path[len - 1];
    const int parent = len >= 2 ? path[len - 2] : 0;
    src_idx[g] = g * slots_per_game + (len >= 2 ? v.hslot[base + parent] : 0);
    dst_idx[g] = g * slots_per_game + dst_slot;
    action_ids[g] = v.rec[base + leaf].action;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
Pool::~Pool()
{
    if (own_stream_ && stream_) { (void)hipStreamDestroy(stream_); }
}

#define MZ_ALLOC(buf, n) \
    if (!(buf).alloc(n)) { setError("pool: allocation of %zu elements failed (%s)", size_t(n), #buf); return MZ_ERR_DEVICE; }

int Pool::init(int device, int games, int nodes_per_game, int action_size, const mz_search_cfg& cfg, hipStream_t shared_stream)
{
    if (games <= 0 || nodes_per_game <= 0 || action_size <= 0 || cfg.num_simulation < 0) { setError("pool: bad arguments"); return MZ_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        setError("no such GPU: device %d of %d (libmzgpu has no CPU path)", device, count);
        return MZ_ERR_DEVICE;
    }
    device_ = device;
    cfg_ = cfg;
    MZ_HIP(hipSetDevice(device));
    if (shared_stream) { stream_ = shared_stream; }
    else { MZ_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking)); own_stream_ = true; }

    const size_t G = games, NN = G * nodes_per_game, GA = G * action_size;
    const int max_depth = cfg.num_simulation + 3; // root + one new level per simulation (+1 Gumbel prefix, +1 slack)
    const int bound_cap = cfg.value_rescale ? cfg.num_simulation + 3 : 1;
    MZ_ALLOC(rec_, NN); MZ_ALLOC(f_nodes_, NN * 3); MZ_ALLOC(i_nodes_, NN);
    MZ_ALLOC(game_i_, G * 3 + 1); MZ_ALLOC(game_f_, G * 2);
    MZ_ALLOC(bound_key_, G * bound_cap); MZ_ALLOC(bound_cnt_, G * bound_cap);
    MZ_ALLOC(bias_tab_, cfg.num_simulation + 3); MZ_ALLOC(sqrt_tab_, cfg.num_simulation + 3); MZ_ALLOC(rcp_tab_, cfg.num_simulation + 5);
    v_.games = games; v_.cap = nodes_per_game; v_.A = action_size; v_.max_depth = max_depth;
    float* f = f_nodes_.p;
    v_.rec = rec_.p;
    v_.logit = f; v_.noise = f + NN; v_.value = f + 2 * NN;
    v_.hslot = i_nodes_.p;
    // path arena: [path_len G][path_action G*max_depth][path G*max_depth]
    MZ_ALLOC(d_path_arena_, G + 2 * G * max_depth); MZ_ALLOC(h_path_arena_, G + 2 * G * max_depth);
    {
        int* dp = reinterpret_cast<int*>(d_path_arena_.p);
        int* hp = reinterpret_cast<int*>(h_path_arena_.p);
        v_.num_nodes = game_i_.p; v_.bound_size = game_i_.p + 2 * G;
        v_.path_len = dp; v_.path_action = dp + G; v_.path = dp + G + G * max_depth;
        h_path_len_ = {hp, G}; h_path_action_ = {hp + G, G * max_depth}; h_path_ = {hp + G + G * max_depth, G * max_depth};
    }
    v_.bound_key = bound_key_.p; v_.bound_cnt = bound_cnt_.p; v_.bound_lo = game_f_.p; v_.bound_hi = game_f_.p + G; v_.bound_cap = bound_cap;
    v_.gamma = cfg.reward_discount; v_.value_rescale = cfg.value_rescale; v_.flipping_player = cfg.flipping_player; v_.atari_init_q = cfg.atari_init_q;

    // PUCT tables (ref mcts.cpp:57-58): float puct_bias = init + log((1 + N + base) / base)  [log = double libm];  sqrt(N) in double
    std::vector<float> bias(cfg.num_simulation + 3);
    std::vector<double> sq(cfg.num_simulation + 3);
    for (int N = 0; N < cfg.num_simulation + 3; ++N) {
        const float ratio = (1 + N + cfg.puct_base) / cfg.puct_base;
        bias[N] = cfg.puct_init + ::log(static_cast<double>(ratio));
        sq[N] = ::sqrt(static_cast<double>(N));
    }
    MZ_HIP(hipMemcpy(bias_tab_.p, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    MZ_HIP(hipMemcpy(sqrt_tab_.p, sq.data(), sq.size() * sizeof(double), hipMemcpyHostToDevice));
    v_.bias_tab = bias_tab_.p;
    v_.sqrt_tab = sqrt_tab_.p;
    std::vector<double> rcp(cfg.num_simulation + 5, 0.0); // correctly rounded reciprocals of the visit counts (host IEEE division)
    for (size_t i = 1; i < rcp.size(); ++i) { rcp[i] = 1.0 / static_cast<double>(i); }
    MZ_HIP(hipMemcpy(rcp_tab_.p, rcp.data(), rcp.size() * sizeof(double), hipMemcpyHostToDevice));
    v_.rcp_tab = rcp_tab_.p;
    MZ_HIP(hipMemset(game_i_.p, 0, game_i_.n * sizeof(int)));

    // staging
    MZ_ALLOC(h_cand_arena_, 4 * G + 3 * GA); MZ_ALLOC(d_cand_arena_, 4 * G + 3 * GA);
    {
        uint32_t *h = h_cand_arena_.p, *d = d_cand_arena_.p;
        h_cand_count_ = {reinterpret_cast<int*>(h), G}; d_cand_count_ = {reinterpret_cast<int*>(d), G};
        h_cand_player_ = {reinterpret_cast<int*>(h + G), G}; d_cand_player_ = {reinterpret_cast<int*>(d + G), G};
        h_value_ = {reinterpret_cast<float*>(h + 2 * G), G}; d_value_ = {reinterpret_cast<float*>(d + 2 * G), G};
        h_reward_ = {reinterpret_cast<float*>(h + 3 * G), G}; d_reward_ = {reinterpret_cast<float*>(d + 3 * G), G};
        h_cand_action_ = {reinterpret_cast<int*>(h + 4 * G), GA}; d_cand_action_ = {reinterpret_cast<int*>(d + 4 * G), GA};
        h_cand_policy_ = {reinterpret_cast<float*>(h + 4 * G + GA), GA}; d_cand_policy_ = {reinterpret_cast<float*>(d + 4 * G + GA), GA};
        h_cand_logit_ = {reinterpret_cast<float*>(h + 4 * G + 2 * GA), GA}; d_cand_logit_ = {reinterpret_cast<float*>(d + 4 * G + 2 * GA), GA};
    }
    MZ_ALLOC(h_start_, G); MZ_ALLOC(d_start_, G); MZ_ALLOC(d_mask_, G);
    MZ_ALLOC(h_flag_, 16);
    h_flag_.p[0] = 0;
    MZ_ALLOC(d_rr_f_, 7 * GA + 5 * G); MZ_ALLOC(d_rr_i_, G + GA + G); MZ_ALLOC(h_rr_f_, 7 * GA + 5 * G); MZ_ALLOC(h_rr_i_, G + GA + G);
    std::vector<int> rp(games, 2);
    return resetSearch(nullptr, rp.data());
}

int Pool::signalAsync(int value)
{
    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, stream_, h_flag_.p, value);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Pool::waitSignal(int value)
{
    volatile int* f = h_flag_.p;
    for (int spin = 0; *f != value; ++spin) {
        __builtin_ia32_pause();
        if (spin > 600000) { // ~20 ms without the signal: let the runtime report what happened
            MZ_HIP(hipStreamSynchronize(stream_));
            if (*f != value) { setError("completion signal %d never arrived (flag = %d)", value, *f); return MZ_ERR_DEVICE; }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return MZ_OK;
}

int Pool::checkError()
{
    int e = 0;
    MZ_HIP(hipMemcpyAsync(&e, game_i_.p + size_t(v_.games) * 3, sizeof(int), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipStreamSynchronize(stream_));
    if (e) {
        (void)hipMemsetAsync(game_i_.p + size_t(v_.games) * 3, 0, sizeof(int), stream_);
        if (e >= 90 && e < 100) { // sim_cluster.h: 90 layer exchange, 91 placement (members on different XCDs), 92 command, 93 results, 94 octet exchange
            setError("simulation kernel: a workgroup of a game's cluster did not arrive (wait %d timed out); the search state of this move is lost", e);
            return MZ_ERR_DEVICE;
        }
        setError("search pool capacity exceeded (nodes_per_game = %d)", v_.cap);
        return e;
    }
    return MZ_OK;
}

int Pool::resetSearch(const uint8_t* mask, const int* root_player)
{
    if (!root_player) { setError("reset_search: root_player is NULL"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    const int G = v_.games;
    // d_start_ doubles as the root-player staging; both uploads complete before the kernel runs (same stream)
    for (int g = 0; g < G; ++g) { h_start_.p[g] = root_player[g]; h_cand_player_.p[g] = mask ? (mask[g] ? 1 : 0) : 1; }
    MZ_HIP(hipMemcpyAsync(d_start_.p, h_start_.p, G * sizeof(int), hipMemcpyHostToDevice, stream_));
    MZ_HIP(hipMemcpyAsync(d_mask_.p, h_cand_player_.p, G * sizeof(int), hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(reset_kernel, dim3((G + 255) / 256), dim3(256), 0, stream_, v_, d_mask_.p, d_start_.p);
    MZ_HIP(hipGetLastError());
    MZ_HIP(hipStreamSynchronize(stream_)); // the pinned mirrors are reused by the caller right after
    return MZ_OK;
}

int Pool::selectAsync(const int* d_start_node)
{
    hipLaunchKernelGGL(select_kernel, dim3(v_.games), dim3(64), 0, stream_, v_, d_start_node);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Pool::select(const int* start_node, int* path_len, int* paths, int* path_action)
{
    MZ_HIP(hipSetDevice(device_));
    const size_t G = v_.games;
    const int* d_start = nullptr;
    if (start_node) {
        memcpy(h_start_.p, start_node, G * sizeof(int));
        MZ_HIP(hipMemcpyAsync(d_start_.p, h_start_.p, G * sizeof(int), hipMemcpyHostToDevice, stream_));
        d_start = d_start_.p;
    }
    int rc = selectAsync(d_start);
    if (rc) { return rc; }
    MZ_HIP(hipMemcpyAsync(h_path_arena_.p, d_path_arena_.p, h_path_arena_.n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipStreamSynchronize(stream_));
    if (path_len) { memcpy(path_len, h_path_len_.p, G * sizeof(int)); }
    if (paths) { memcpy(paths, h_path_.p, G * v_.max_depth * sizeof(int)); }
    if (path_action) { memcpy(path_action, h_path_action_.p, G * v_.max_depth * sizeof(int)); }
    return MZ_OK;
}

int Pool::expandBackupAsync(int hslot, bool from_host)
{
    const size_t lds = v_.value_rescale ? size_t(v_.bound_cap) * 8 : 0;
    if (from_host) { // zero-copy: the kernel reads the pinned candidate arena over PCIe (<= 1 KB per game)
        hipLaunchKernelGGL(expand_backup_kernel, dim3(v_.games), dim3(64), lds, stream_, v_, h_cand_count_.p, h_cand_action_.p, h_cand_policy_.p,
                           h_cand_logit_.p, h_cand_player_.p, h_value_.p, h_reward_.p, hslot, game_i_.p + size_t(v_.games) * 3);
        MZ_HIP(hipGetLastError());
        return MZ_OK;
    }
    hipLaunchKernelGGL(expand_backup_kernel, dim3(v_.games), dim3(64), lds, stream_, v_, d_cand_count_.p, d_cand_action_.p, d_cand_policy_.p,
                       d_cand_logit_.p, d_cand_player_.p, d_value_.p, d_reward_.p, hslot, game_i_.p + size_t(v_.games) * 3);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Pool::expandBackupStaged(int hslot)
{
    if (zero_copy_) { return expandBackupAsync(hslot, true); }
    MZ_HIP(hipMemcpyAsync(d_cand_arena_.p, h_cand_arena_.p, h_cand_arena_.n * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
    return expandBackupAsync(hslot);
}

int Pool::expandBackup(const int* cand_count, const int* cand_action, const float* cand_policy, const float* cand_logit, const int* cand_player,
                       const float* value, const float* reward)
{
    if (!cand_count || !cand_action || !cand_policy || !cand_logit || !cand_player || !value) { setError("expand_backup: NULL argument"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    const size_t G = v_.games, GA = G * v_.A;
    for (size_t g = 0; g < G; ++g) {
        if (cand_count[g] < 0 || cand_count[g] > v_.A) { setError("expand_backup: cand_count[%zu] = %d out of range", g, cand_count[g]); return MZ_ERR_ARG; }
    }
    memcpy(h_cand_count_.p, cand_count, G * sizeof(int));
    memcpy(h_cand_action_.p, cand_action, GA * sizeof(int));
    memcpy(h_cand_policy_.p, cand_policy, GA * sizeof(float));
    memcpy(h_cand_logit_.p, cand_logit, GA * sizeof(float));
    memcpy(h_cand_player_.p, cand_player, G * sizeof(int));
    memcpy(h_value_.p, value, G * sizeof(float));
    if (reward) { memcpy(h_reward_.p, reward, G * sizeof(float)); } else { memset(h_reward_.p, 0, G * sizeof(float)); }
    int rc = expandBackupStaged(hslot_next_);
    if (rc) { return rc; }
    return checkError();
}

int Pool::rootSetNoise(const uint8_t* mask, const float* policy, const float* logit, const float* noise)
{
    if (!policy || !logit || !noise) { setError("root_set_noise: NULL argument"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    const size_t G = v_.games, GA = G * v_.A;
    for (size_t g = 0; g < G; ++g) { h_cand_count_.p[g] = mask ? (mask[g] ? 1 : 0) : 1; }
    memcpy(h_cand_policy_.p, policy, GA * sizeof(float));
    memcpy(h_cand_logit_.p, logit, GA * sizeof(float));
    // noise travels through the (otherwise idle) root-read float staging
    memcpy(h_rr_f_.p, noise, GA * sizeof(float));
    MZ_HIP(hipMemcpyAsync(d_mask_.p, h_cand_count_.p, G * sizeof(int), hipMemcpyHostToDevice, stream_));
    MZ_HIP(hipMemcpyAsync(d_cand_policy_.p, h_cand_policy_.p, GA * sizeof(float), hipMemcpyHostToDevice, stream_));
    MZ_HIP(hipMemcpyAsync(d_cand_logit_.p, h_cand_logit_.p, GA * sizeof(float), hipMemcpyHostToDevice, stream_));
    MZ_HIP(hipMemcpyAsync(d_rr_f_.p, h_rr_f_.p, GA * sizeof(float), hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(root_set_noise_kernel, dim3(v_.games), dim3(64), 0, stream_, v_, d_mask_.p, d_cand_policy_.p, d_cand_logit_.p, d_rr_f_.p);
    MZ_HIP(hipGetLastError());
    MZ_HIP(hipStreamSynchronize(stream_));
    return MZ_OK;
}

// the kernel (+ copies) of rootRead, queued on the stream without waiting: the worker puts it right behind the launch that completes a move's search, so that the
// statistics are on their way while the host still waits for that launch (a launch + a wake-up less on the critical path between two moves)
int Pool::rootReadLaunch()
{
    MZ_HIP(hipSetDevice(device_));
    const size_t G = v_.games, GA = G * v_.A;
    // A small pool's statistics (64 Atari games: 37 KB) are written by the kernel straight into the pinned host buffers — a kernel and one wait instead of a
    // kernel, two copies and a wait on the host's critical path between two moves; a large pool's (256 Go games: 0.7 MB) take the copy engine
    if ((8 * GA + 7 * G) * sizeof(float) <= size_t(128) * 1024) {
        hipLaunchKernelGGL(root_read_kernel, dim3(v_.games), dim3(64), 0, stream_, v_, h_rr_f_.p, h_rr_i_.p);
        MZ_HIP(hipGetLastError());
    } else {
        hipLaunchKernelGGL(root_read_kernel, dim3(v_.games), dim3(64), 0, stream_, v_, d_rr_f_.p, d_rr_i_.p);
        MZ_HIP(hipGetLastError());
        MZ_HIP(hipMemcpyAsync(h_rr_f_.p, d_rr_f_.p, (7 * GA + 5 * G) * sizeof(float), hipMemcpyDeviceToHost, stream_));
        MZ_HIP(hipMemcpyAsync(h_rr_i_.p, d_rr_i_.p, (G + GA + G) * sizeof(int), hipMemcpyDeviceToHost, stream_));
    }
    return MZ_OK;
}

int Pool::rootRead(int* num_children, int* action, float* count, float* mean, float* policy, float* logit, float* noise, float* value, float* reward,
                   float* root_count, float* root_mean, float* root_value, float* bound_lo, float* bound_hi, int* bound_size, bool launched_ahead)
{
    MZ_HIP(hipSetDevice(device_));
    const size_t G = v_.games, GA = G * v_.A;
    if (!launched_ahead) { int rc = rootReadLaunch(); if (rc) { return rc; } }
    MZ_HIP(hipStreamSynchronize(stream_));
    const float* f = h_rr_f_.p;
    const int* iv = h_rr_i_.p;
    float* outs[7] = {count, mean, policy, logit, noise, value, reward};
    for (int k = 0; k < 7; ++k) { if (outs[k]) { memcpy(outs[k], f + k * GA, GA * sizeof(float)); } }
    float* pg[5] = {root_count, root_mean, root_value, bound_lo, bound_hi};
    for (int k = 0; k < 5; ++k) { if (pg[k]) { memcpy(pg[k], f + 7 * GA + k * G, G * sizeof(float)); } }
    if (num_children) { memcpy(num_children, iv, G * sizeof(int)); }
    if (action) { memcpy(action, iv + G, GA * sizeof(int)); }
    if (bound_size) { memcpy(bound_size, iv + G + GA, G * sizeof(int)); }
    return MZ_OK;
}

int Pool::hiddenIndexAsync(int slots_per_game, int dst_slot, int* d_src_idx, int* d_dst_idx, int* d_action_ids)
{
    hipLaunchKernelGGL(hidden_index_kernel, dim3((v_.games + 255) / 256), dim3(256), 0, stream_, v_, slots_per_game, dst_slot, d_src_idx, d_dst_idx,
                       d_action_ids);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Pool::numNodes(int game)
{
    if (game < 0 || game >= v_.games) { setError("num_nodes: bad game index"); return MZ_ERR_ARG; }
    int n = 0;
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipMemcpyAsync(&n, v_.num_nodes + game, sizeof(int), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipStreamSynchronize(stream_));
    return n;
}

int Pool::readNodes(int game, int n, int* action, int* player, int* num_children, int* first_child, float* mean, float* count, float* policy,
                    float* logit, float* noise, float* value, float* reward)
{
    if (game < 0 || game >= v_.games || n < 0 || n > v_.cap) { setError("read_nodes: bad arguments"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipStreamSynchronize(stream_));
    const size_t base = size_t(game) * v_.cap;
    std::vector<NodeRec> recs(n);
    MZ_HIP(hipMemcpy(recs.data(), v_.rec + base, size_t(n) * sizeof(NodeRec), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        const NodeRec& r = recs[i];
        if (action) { action[i] = r.action; }
        if (player) { player[i] = r.players & 0xFF; }
        if (num_children) { num_children[i] = r.num_children; }
        if (first_child) { first_child[i] = r.first_child; }
        if (mean) { mean[i] = r.mean; }
        if (count) { count[i] = r.count; }
        if (policy) { policy[i] = r.policy; }
        if (reward) { reward[i] = r.reward; }
    }
    auto cp = [&](void* dst, const void* src, size_t bytes) { return dst ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) : hipSuccess; };
    MZ_HIP(cp(logit, v_.logit + base, n * sizeof(float)));
    MZ_HIP(cp(noise, v_.noise + base, n * sizeof(float)));
    MZ_HIP(cp(value, v_.value + base, n * sizeof(float)));
    return MZ_OK;
}

} // namespace mz
