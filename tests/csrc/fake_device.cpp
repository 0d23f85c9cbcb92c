// TEST INFRASTRUCTURE ONLY — never part of libmzgpu.so, never shipped, never measured.
//
// A GPU-less stand-in for the DEVICE side of the worker so that the worker's HOST half (minizero_amd/csrc/worker.cpp: the RNG-ordered per-move
// logic, the records, the spin-wait thread pool, the RNG streams' sinks, the command handling) can run under ThreadSanitizer / AddressSanitizer
// on a machine without a GPU (`pytest -m "not gpu"`, tests/test_sanitizers.py) — the seam is what `nm -u worker.o` lists: ~25 HIP runtime
// calls and ~45 methods of mz::Net / mz::Pool / mz::GoDevice.
//
// What stands behind the seam here is the CHECKER, not a second implementation: the tree is the oracle's MCTS (oracle/o_mcts.cpp through its
// `mzo_tree_*` entry points), the network the oracle's forward (`mzo_net_*`).  "Device memory" is host memory, a "stream" runs every operation
// at once.  Only the lock-step mode with the rules on the host is served (hasSimKernel* / hasFusedTower say no; mz_device_env=false is
// required) for AlphaZero and MuZero networks (`muzero_atari` with float observations, mz_raw_observations=false); the device rules and every simulation kernel return an error.
// The product has no such path: libmzgpu.so without a GPU fails with MZ_ERR_DEVICE (tests/test_capi.py).
#include "../../minizero_amd/csrc/net.h"
#include "../../minizero_amd/csrc/pool.h"
#include "../../minizero_amd/csrc/go_dev.h"
#include "../../minizero_amd/csrc/gumbel.h"
#include "../../minizero_amd/csrc/loader_dev.h"
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

// ---- the oracle's C entry points (oracle/o_capi.cpp) ----
extern "C" {
void* mzo_net_create(const mz_net_desc* d, const float* raw, long n); // (mzo::NetDesc has the layout of mz_net_desc: tests/test_capi.py)
void mzo_net_destroy(void* net);
void mzo_net_forward_az(void* net, const float* feat, int B, float* policy, float* logit, float* value);
void mzo_net_initial(void* net, const float* feat, int B, float* policy, float* logit, float* value, float* hidden);
void mzo_net_recurrent(void* net, const float* hidden_in, const float* action, int B, float* policy, float* logit, float* value, float* reward, float* hidden_out);
void* mzo_tree_create(const char* conf, long tree_node_size);
void mzo_tree_destroy(void* t);
void mzo_tree_reset(void* t, int root_player);
int mzo_tree_select(void* t, int start, int* path_out, int cap);
void mzo_tree_expand_backup(void* t, int k, const int* action_ids, int player, const float* policy, const float* logit, float value, float reward);
void mzo_tree_set_child_policy(void* t, int node, float policy, float logit, float noise);
int mzo_tree_num_nodes(void* t);
void mzo_tree_dump(void* t, int n, int* action, int* player, int* num_children, int* first_child, float* mean, float* count, float* policy, float* logit,
                   float* noise, float* value, float* reward);
int mzo_tree_value_bound(void* t, float* lo, float* hi);
}

// ------------------------------------------------------------------------------------------------
// HIP runtime: host memory, immediate execution
// ------------------------------------------------------------------------------------------------
extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake device"; }
#ifndef MZ_FAKE_WITH_CAPI // (with capi.cpp linked in — the `-mode sp` executable over the facades — these come from there)
int mz_device_count(void) { return 1; }
const char* mz_last_error(void) { return mz::lastError(); }
#endif
}

namespace mz {

#ifndef MZ_FAKE_WITH_CAPI
// ---- what capi.cpp provides in the product ----
static thread_local char g_err[1024] = "";
void setError(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* lastError() { return g_err; }
bool readWeightFile(const std::string& path, mz_net_desc*, std::vector<float>*) { setError("fake device: no weight files (%s)", path.c_str()); return false; }
#else
// ---- what capi.cpp asks of the device beyond the worker's seam: the stand-alone network / pool entry points are not served ----
int Net::forwardAZ_any(const float*, int, float*, float*, float*, int) { setError("fake device: mz_net_forward_az is not served"); return MZ_ERR_DEVICE; }
int Net::initial_any(const float*, int, float*, float*, float*, float*, int) { setError("fake device: mz_net_initial is not served"); return MZ_ERR_DEVICE; }
int Net::recurrent_any(const float*, const float*, int, float*, float*, float*, float*, float*, int) { setError("fake device: mz_net_recurrent is not served"); return MZ_ERR_DEVICE; }
int Net::timeForward(int, int, float*, float*, double*) { setError("fake device: no timing"); return MZ_ERR_DEVICE; }
int Net::timeTowerConv(int, int, float*, double*, double*) { setError("fake device: no timing"); return MZ_ERR_DEVICE; }
int Pool::select(const int*, int*, int*, int*) { setError("fake device: mz_pool_select is not served"); return MZ_ERR_DEVICE; }
int Pool::expandBackup(const int*, const int*, const float*, const float*, const int*, const float*, const float*) { setError("fake device: mz_pool_expand_backup is not served"); return MZ_ERR_DEVICE; }
int Pool::numNodes(int) { setError("fake device: mz_pool_num_nodes is not served"); return MZ_ERR_DEVICE; }
int Pool::readNodes(int, int, int*, int*, int*, int*, float*, float*, float*, float*, float*, float*, float*) { setError("fake device: mz_pool_read_nodes is not served"); return MZ_ERR_DEVICE; }
#endif
float invertValueHost(float v) { return v; } // (muzero_atari: the oracle's network hands out DECODED values — muzero_network.h:157-174 — where the HIP heads hand out the transformed expectation)
int invertValuesOnDevice(int, const float*, int, float*) { setError("fake device"); return MZ_ERR_DEVICE; }
int sortCandidatesOnDevice(int, const float*, int, int*) { setError("fake device"); return MZ_ERR_DEVICE; }

namespace {
int refuse(const char* what) { setError("fake device: %s is not served (lock-step mode with host rules only)", what); return MZ_ERR_DEVICE; }

struct NetState { void* onet = nullptr; };
struct PoolState {
    std::vector<void*> trees;
    std::vector<std::vector<int>> action_of, hslot_of; // per game, per node
    std::vector<int> owned;
};
std::mutex g_mu;
std::mutex g_oracle_net_mu; // the oracle's forward runs on ONE process-wide parallel-for (oracle/o_nn.cpp): one caller at a time — several workers (logical devices) take turns
std::map<const Net*, NetState> g_nets;
std::map<const Pool*, PoolState> g_pools;
NetState& stateOf(const Net* n) { std::lock_guard<std::mutex> l(g_mu); return g_nets[n]; }
PoolState& stateOf(const Pool* p) { std::lock_guard<std::mutex> l(g_mu); return g_pools[p]; }
} // namespace

// ------------------------------------------------------------------------------------------------
// Net: the oracle's forward
// ------------------------------------------------------------------------------------------------
Net::~Net()
{
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_nets.find(this);
    if (it != g_nets.end()) { if (it->second.onet) { mzo_net_destroy(it->second.onet); } g_nets.erase(it); }
    if (own_stream_ && stream_) { (void)hipStreamDestroy(stream_); }
}
int Net::init(int device, const mz_net_desc& d, const float* raw, size_t n)
{
    desc_ = d; // (muzero_atari too: the oracle's forward returns value / reward already decoded, and invertValueHost above is the identity to match)
    device_ = device;
    MZ_HIP(hipStreamCreateWithFlags(&stream_, 0));
    own_stream_ = true;
    return reload(raw, n);
}
int Net::reload(const float* raw, size_t n)
{
    void* nn = mzo_net_create(&desc_, raw, static_cast<long>(n));
    if (!nn) { setError("fake device: the oracle refused the parameters (%zu values)", n); return MZ_ERR_ARG; }
    NetState& s = stateOf(this);
    if (s.onet) { mzo_net_destroy(s.onet); }
    s.onet = nn;
    return MZ_OK;
}
int Net::setPrecision(int mode) { return mode == 0 ? MZ_OK : refuse("bf16x3"); }
bool Net::hasFusedTower() { return false; }
bool Net::hasSimKernel(int, int, int) const { return false; }
bool Net::hasSimKernelMz(int) const { return false; }
bool Net::hasPreBoard() const { return false; }
int Net::forwardAZ(const float* d_feat, int B, float* d_policy, float* d_logit, float* d_value, bool in_bits)
{
    const int C = desc_.num_input_channels, P = desc_.input_channel_height * desc_.input_channel_width, W32 = (P + 31) / 32;
    std::vector<float> planes;
    const float* feat = d_feat;
    if (in_bits) { // GameEnv::featureBits layout: [sample][channel][W32 words], bit p of a plane
        const uint32_t* bits = reinterpret_cast<const uint32_t*>(d_feat);
        planes.resize(size_t(B) * C * P);
        for (int b = 0; b < B; ++b) {
            for (int c = 0; c < C; ++c) {
                for (int p = 0; p < P; ++p) { planes[(size_t(b) * C + c) * P + p] = ((bits[(size_t(b) * C + c) * W32 + (p >> 5)] >> (p & 31)) & 1u) ? 1.0f : 0.0f; }
            }
        }
        feat = planes.data();
    }
    { std::lock_guard<std::mutex> l(g_oracle_net_mu); mzo_net_forward_az(stateOf(this).onet, feat, B, d_policy, d_logit, d_value); }
    return MZ_OK;
}
int Net::initial(const float* d_feat, int B, float* d_policy, float* d_logit, float* d_value, float* d_hidden, const int* d_dst_idx)
{
    const size_t hs = size_t(hiddenSize());
    std::vector<float> hidden(size_t(B) * hs);
    { std::lock_guard<std::mutex> l(g_oracle_net_mu); mzo_net_initial(stateOf(this).onet, d_feat, B, d_policy, d_logit, d_value, hidden.data()); }
    for (int b = 0; b < B; ++b) { memcpy(d_hidden + size_t(d_dst_idx ? d_dst_idx[b] : b) * hs, hidden.data() + size_t(b) * hs, hs * sizeof(float)); }
    return MZ_OK;
}
int Net::recurrent(const float* d_hidden_src, const int* d_src_idx, const float* d_action_planes, const int* d_action_ids, int B, float* d_policy,
                   float* d_logit, float* d_value, float* d_reward, float* d_hidden_dst, const int* d_dst_idx)
{
    const size_t hs = size_t(hiddenSize());
    const int AC = desc_.num_action_feature_channels, P = this->P();
    std::vector<float> hin(size_t(B) * hs), hout(size_t(B) * hs), act(size_t(B) * AC * P, 0.0f);
    for (int b = 0; b < B; ++b) {
        memcpy(hin.data() + size_t(b) * hs, d_hidden_src + size_t(d_src_idx ? d_src_idx[b] : b) * hs, hs * sizeof(float));
        if (d_action_planes) { memcpy(act.data() + size_t(b) * AC * P, d_action_planes + size_t(b) * AC * P, size_t(AC) * P * sizeof(float)); }
        else { // net.hip build_recurrent_input: one plane = one-hot at the action's point (pass: none); several planes = the action's plane all ones
            for (int i = 0; i < AC * P; ++i) { act[size_t(b) * AC * P + i] = (AC == 1) ? (i == d_action_ids[b] ? 1.0f : 0.0f) : ((i / P) == d_action_ids[b] ? 1.0f : 0.0f); }
        }
    }
    { std::lock_guard<std::mutex> l(g_oracle_net_mu); mzo_net_recurrent(stateOf(this).onet, hin.data(), act.data(), B, d_policy, d_logit, d_value, d_reward, hout.data()); }
    for (int b = 0; b < B; ++b) { memcpy(d_hidden_dst + size_t(d_dst_idx ? d_dst_idx[b] : b) * hs, hout.data() + size_t(b) * hs, hs * sizeof(float)); }
    return MZ_OK;
}
int Net::simLaunch(Pool&, const GoDevView&, float*, float*, float*, const uint8_t*, int, int, bool*, const float*, float, int, const GumbelView*, int*, bool) { return refuse("sim_kernel"); }
int Net::simLaunchMz(Pool&, float*, int, const unsigned*, const unsigned long long*, const int*, int, float*, float*, float*, float*, int, int, bool*, const float*, float, int,
                     const GumbelView*, int*, bool, bool, int, bool, const SimMzMode&) { return refuse("sim_kernel_mz"); }
int Net::simPreEvalMz(int, int, int, int, int, bool*, bool, bool) { return refuse("sim_pre_kernel_mz"); }
int Net::simPreEvalBatchMz(int, int, int, int, int, bool*, int, bool) { return refuse("the batched rounds"); }
int Net::simRootNoiseMz(int) { return refuse("sim_root_noise_kernel"); }
int Net::simPreStats(unsigned* hits, unsigned* evals, unsigned* alt) { *hits = *evals = *alt = 0; return MZ_OK; }
int Net::simPreCountersAsync(unsigned*) { return MZ_OK; }
int Net::expandAtariFeatures(const uint8_t*, int, int, float*) { return refuse("muzero_atari"); }
int Net::shiftExpandAtariFeatures(const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*, int, int, float*) { return refuse("muzero_atari"); }

// ------------------------------------------------------------------------------------------------
// Pool: one oracle tree per game behind the staging layout of pool.hip
// ------------------------------------------------------------------------------------------------
Pool::~Pool()
{
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_pools.find(this);
    if (it != g_pools.end()) { for (void* t : it->second.trees) { mzo_tree_destroy(t); } g_pools.erase(it); }
    if (own_stream_ && stream_) { (void)hipStreamDestroy(stream_); }
}
int Pool::init(int device, int games, int nodes_per_game, int action_size, const mz_search_cfg& cfg, hipStream_t shared_stream)
{
    device_ = device;
    cfg_ = cfg;
    if (shared_stream) { stream_ = shared_stream; } else { MZ_HIP(hipStreamCreateWithFlags(&stream_, 0)); own_stream_ = true; }
    const size_t G = games, GA = G * action_size;
    const int max_depth = cfg.num_simulation + 3;
    v_ = PoolView{};
    v_.games = games; v_.cap = nodes_per_game; v_.A = action_size; v_.max_depth = max_depth;
    v_.bound_cap = cfg.value_rescale ? cfg.num_simulation + 3 : 1;
    if (!game_i_.alloc(G * 3 + 1) || !d_path_arena_.alloc(G + 2 * G * max_depth) || !h_path_arena_.alloc(G + 2 * G * max_depth) || !h_cand_arena_.alloc(4 * G + 3 * GA) ||
        !d_cand_arena_.alloc(4 * G + 3 * GA) || !h_start_.alloc(G) || !d_start_.alloc(G) || !d_mask_.alloc(G) || !h_flag_.alloc(16) || !d_rr_f_.alloc(7 * GA + 5 * G) ||
        !d_rr_i_.alloc(G + GA + G) || !h_rr_f_.alloc(7 * GA + 5 * G) || !h_rr_i_.alloc(G + GA + G)) {
        setError("fake device: allocation failed");
        return MZ_ERR_DEVICE;
    }
    { // the layout of pool.hip Pool::init (the worker indexes these views directly)
        int* dp = reinterpret_cast<int*>(d_path_arena_.p);
        int* hp = reinterpret_cast<int*>(h_path_arena_.p);
        v_.num_nodes = game_i_.p; v_.bound_size = game_i_.p + 2 * G;
        v_.path_len = dp; v_.path_action = dp + G; v_.path = dp + G + G * max_depth;
        h_path_len_ = {hp, G}; h_path_action_ = {hp + G, G * size_t(max_depth)}; h_path_ = {hp + G + G * max_depth, G * size_t(max_depth)};
        uint32_t *h = h_cand_arena_.p, *d = d_cand_arena_.p;
        h_cand_count_ = {reinterpret_cast<int*>(h), G}; d_cand_count_ = {reinterpret_cast<int*>(d), G};
        h_cand_player_ = {reinterpret_cast<int*>(h + G), G}; d_cand_player_ = {reinterpret_cast<int*>(d + G), G};
        h_value_ = {reinterpret_cast<float*>(h + 2 * G), G}; d_value_ = {reinterpret_cast<float*>(d + 2 * G), G};
        h_reward_ = {reinterpret_cast<float*>(h + 3 * G), G}; d_reward_ = {reinterpret_cast<float*>(d + 3 * G), G};
        h_cand_action_ = {reinterpret_cast<int*>(h + 4 * G), GA}; d_cand_action_ = {reinterpret_cast<int*>(d + 4 * G), GA};
        h_cand_policy_ = {reinterpret_cast<float*>(h + 4 * G + GA), GA}; d_cand_policy_ = {reinterpret_cast<float*>(d + 4 * G + GA), GA};
        h_cand_logit_ = {reinterpret_cast<float*>(h + 4 * G + 2 * GA), GA}; d_cand_logit_ = {reinterpret_cast<float*>(d + 4 * G + 2 * GA), GA};
    }
    h_flag_.p[0] = 0;
    char conf[512];
    snprintf(conf, sizeof(conf),
             "actor_num_simulation=%d:actor_mcts_puct_base=%.9g:actor_mcts_puct_init=%.9g:actor_mcts_reward_discount=%.9g:actor_mcts_value_rescale=%s:"
             "actor_mcts_value_flipping_player=%c:atari_init_q=%s",
             cfg.num_simulation, cfg.puct_base, cfg.puct_init, cfg.reward_discount, cfg.value_rescale ? "true" : "false", cfg.flipping_player == 1 ? 'B' : 'W',
             cfg.atari_init_q ? "true" : "false");
    PoolState& s = stateOf(this);
    for (int g = 0; g < games; ++g) {
        void* t = mzo_tree_create(conf, nodes_per_game - 1);
        if (!t) { setError("fake device: the oracle refused the search configuration %s", conf); return MZ_ERR_ARG; }
        s.trees.push_back(t);
    }
    s.action_of.assign(games, {});
    s.hslot_of.assign(games, {});
    std::vector<int> rp(games, 2);
    return resetSearch(nullptr, rp.data());
}
int Pool::resetSearch(const uint8_t* mask, const int* root_player)
{
    PoolState& s = stateOf(this);
    for (int g = 0; g < v_.games; ++g) {
        if (mask && !mask[g]) { continue; }
        mzo_tree_reset(s.trees[g], root_player[g]);
        s.action_of[g].assign(1, -1);
        s.hslot_of[g].assign(1, -1);
        v_.num_nodes[g] = 1;
        v_.path_len[g] = 0;
    }
    return MZ_OK;
}
int Pool::selectAsync(const int* d_start_node)
{
    PoolState& s = stateOf(this);
    std::vector<int> path(v_.max_depth + 4);
    for (int g = 0; g < v_.games; ++g) {
        const int start = (d_start_node && d_start_node[g] > 0) ? d_start_node[g] : -1;
        const int len = mzo_tree_select(s.trees[g], start, path.data(), static_cast<int>(path.size()));
        if (len > v_.max_depth) { setError("fake device: a path of %d nodes", len); return MZ_ERR_CAPACITY; }
        v_.path_len[g] = len;
        for (int d = 0; d < len; ++d) {
            v_.path[size_t(g) * v_.max_depth + d] = path[d];
            v_.path_action[size_t(g) * v_.max_depth + d] = s.action_of[g][path[d]];
        }
        if (v_.host_path_len) { // the zero-copy mirrors select_kernel writes
            v_.host_path_len[g] = len;
            for (int d = 0; d < len; ++d) { v_.host_path_action[size_t(g) * v_.max_depth + d] = s.action_of[g][path[d]]; }
        }
    }
    return MZ_OK;
}
int Pool::expandBackupAsync(int hslot, bool from_host)
{
    PoolState& s = stateOf(this);
    const int* cnt = from_host ? h_cand_count_.p : d_cand_count_.p;
    const int* act = from_host ? h_cand_action_.p : d_cand_action_.p;
    const int* ply = from_host ? h_cand_player_.p : d_cand_player_.p;
    const float* pol = from_host ? h_cand_policy_.p : d_cand_policy_.p;
    const float* lgt = from_host ? h_cand_logit_.p : d_cand_logit_.p;
    const float* val = from_host ? h_value_.p : d_value_.p;
    const float* rew = from_host ? h_reward_.p : d_reward_.p;
    for (int g = 0; g < v_.games; ++g) {
        const int len = v_.path_len[g];
        if (len < 1) { continue; }
        const int leaf = v_.path[size_t(g) * v_.max_depth + len - 1], k = cnt[g];
        if (k > 0 && v_.num_nodes[g] + k > v_.cap) { game_i_.p[size_t(v_.games) * 3] = MZ_ERR_CAPACITY; continue; }
        mzo_tree_expand_backup(s.trees[g], k, act + size_t(g) * v_.A, ply[g], pol + size_t(g) * v_.A, lgt + size_t(g) * v_.A, val[g], rew[g]);
        for (int i = 0; i < k; ++i) { s.action_of[g].push_back(act[size_t(g) * v_.A + i]); s.hslot_of[g].push_back(-1); }
        v_.num_nodes[g] += k;
        if (hslot >= 0) { s.hslot_of[g][leaf] = hslot; }
        if (mzo_tree_num_nodes(s.trees[g]) != v_.num_nodes[g]) { setError("fake device: node count drifted from the oracle tree's"); return MZ_ERR_STATE; }
    }
    return MZ_OK;
}
int Pool::expandBackupStaged(int hslot)
{
    if (zero_copy_) { return expandBackupAsync(hslot, true); }
    memcpy(d_cand_arena_.p, h_cand_arena_.p, h_cand_arena_.n * sizeof(uint32_t));
    return expandBackupAsync(hslot, false);
}
int Pool::rootSetNoise(const uint8_t* mask, const float* policy, const float* logit, const float* noise)
{
    PoolState& s = stateOf(this);
    std::vector<int> nc(1), dummy_i(1);
    for (int g = 0; g < v_.games; ++g) {
        if (mask && !mask[g]) { continue; }
        int action, player, num_children, first_child;
        float f[7];
        mzo_tree_dump(s.trees[g], 1, &action, &player, &num_children, &first_child, f, f + 1, f + 2, f + 3, f + 4, f + 5, f + 6);
        for (int i = 0; i < num_children; ++i) {
            const size_t c = size_t(g) * v_.A + i;
            mzo_tree_set_child_policy(s.trees[g], first_child + i, policy[c], logit[c], noise[c]);
        }
    }
    return MZ_OK;
}
int Pool::rootReadLaunch()
{
    PoolState& s = stateOf(this);
    const size_t G = v_.games, GA = G * v_.A;
    float* f = h_rr_f_.p;
    int* iv = h_rr_i_.p;
    std::vector<int> action(v_.A + 1), player(v_.A + 1), nch(v_.A + 1), fch(v_.A + 1);
    std::vector<float> mean(v_.A + 1), count(v_.A + 1), policy(v_.A + 1), logit(v_.A + 1), noise(v_.A + 1), value(v_.A + 1), reward(v_.A + 1);
    for (size_t g = 0; g < G; ++g) {
        mzo_tree_dump(s.trees[g], 1, action.data(), player.data(), nch.data(), fch.data(), mean.data(), count.data(), policy.data(), logit.data(), noise.data(), value.data(),
                      reward.data());
        const int nc = nch[0];
        float* pg = f + 7 * GA;
        pg[0 * G + g] = count[0]; pg[1 * G + g] = mean[0]; pg[2 * G + g] = value[0];
        float lo = 0.0f, hi = 0.0f;
        const int bsize = mzo_tree_value_bound(s.trees[g], &lo, &hi);
        pg[3 * G + g] = lo; pg[4 * G + g] = hi;
        iv[g] = nc;
        iv[G + GA + g] = bsize;
        if (nc > 0) {
            if (fch[0] != 1) { setError("fake device: the root's children do not start at node 1"); return MZ_ERR_STATE; }
            mzo_tree_dump(s.trees[g], 1 + nc, action.data(), player.data(), nch.data(), fch.data(), mean.data(), count.data(), policy.data(), logit.data(), noise.data(),
                          value.data(), reward.data());
        }
        for (int i = 0; i < nc; ++i) {
            const size_t c = g * v_.A + i;
            f[0 * GA + c] = count[1 + i]; f[1 * GA + c] = mean[1 + i]; f[2 * GA + c] = policy[1 + i]; f[3 * GA + c] = logit[1 + i];
            f[4 * GA + c] = noise[1 + i]; f[5 * GA + c] = value[1 + i]; f[6 * GA + c] = reward[1 + i];
            iv[G + c] = action[1 + i];
        }
    }
    return MZ_OK;
}
int Pool::rootRead(int* num_children, int* action, float* count, float* mean, float* policy, float* logit, float* noise, float* value, float* reward,
                   float* root_count, float* root_mean, float* root_value, float* bound_lo, float* bound_hi, int* bound_size, bool launched_ahead)
{
    const size_t G = v_.games, GA = G * v_.A;
    if (!launched_ahead) { int rc = rootReadLaunch(); if (rc) { return rc; } }
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
    PoolState& s = stateOf(this);
    for (int g = 0; g < v_.games; ++g) { // pool.hip hidden_index_kernel
        const int len = v_.path_len[g];
        const int* path = v_.path + size_t(g) * v_.max_depth;
        const int leaf = path[len - 1], parent = len >= 2 ? path[len - 2] : 0;
        d_src_idx[g] = g * slots_per_game + (len >= 2 ? s.hslot_of[g][parent] : 0);
        d_dst_idx[g] = g * slots_per_game + dst_slot;
        d_action_ids[g] = s.action_of[g][leaf];
    }
    return MZ_OK;
}
int Pool::checkError()
{
    int& e = game_i_.p[size_t(v_.games) * 3];
    if (e) { e = 0; setError("search pool capacity exceeded (nodes_per_game = %d)", v_.cap); return MZ_ERR_CAPACITY; }
    return MZ_OK;
}
int Pool::signalAsync(int value) { h_flag_.p[0] = value; return MZ_OK; }
int Pool::waitSignal(int value)
{
    if (h_flag_.p[0] != value) { setError("fake device: completion signal %d never arrived (flag = %d)", value, h_flag_.p[0]); return MZ_ERR_DEVICE; }
    return MZ_OK;
}

// the learner-side sampler's device half (loader_kernels.hip): not served — the fuzzers only feed its record parser
int loaderReplayFeatures(GoDevice&, const PoolView&, int, const int*, const uint8_t*, float*, hipStream_t) { return refuse("the sampler's feature replay"); }
int loaderExpandAtari(const uint8_t*, int, float*, hipStream_t) { return refuse("the sampler's screen expansion"); }

// ------------------------------------------------------------------------------------------------
// device rules: not served
// ------------------------------------------------------------------------------------------------
int GoDevice::init(int, int, int, float, int, int, int, hipStream_t, const int* const[8], const int* const[8], const uint64_t*, int, uint64_t) { return refuse("the device rules (set mz_device_env=false)"); }
int GoDevice::uploadRoots() { return refuse("the device rules"); }
int GoDevice::leafAsync(const PoolView&, const RotPack&, int) { return refuse("the device rules"); }
int GoDevice::candAsync(Pool&, const float*, const float*, const float*, const RotPack&) { return refuse("the device rules"); }
int GoDevice::readLeaf(uint32_t*, uint8_t*, int*, float*, int*) { return refuse("the device rules"); }

} // namespace mz
