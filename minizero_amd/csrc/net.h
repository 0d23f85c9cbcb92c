// Network inference path of libmzgpu: BN-folded weights resident in HBM, im2col-free 3x3 convolutions on
// the f32 MFMA pipe with an LDS-staged zero-padded input tile, fused bias(+skip)+ReLU, fused heads.
// Math being reproduced: ref network/py/network_unit.py:6-87, alphazero_network.py:90-113,
// muzero_network.py:137-164.  Arithmetic order is specified in DESIGN.md §"Network numerics".
#pragma once
#include "common.h"
#include <string>

namespace mz {

struct ConvLayer {
    int cin, cin_pad, cout, cout_pad;
    size_t w_off, b_off; // offsets (floats) into Net::params_
    size_t w4_off;       // the same A-fragments, four channel groups interleaved per lane (one dwordx4 load = 4 k-steps): the fused tower's layout
    size_t wq_off;       // ... with the input channels padded to a multiple of 16: [tap][oc-tile][cq chunks][lane][4] (net_wide_body.h); == w4_off when cin_pad % 16 == 0
    int cq;              // dwordx4 chunks (16 input channels) per (tap, oc-tile) in the wq layout
};

struct HeadOffsets {
    int pc; // policy conv channels = ceil(A / (h*w)) (ref network_unit.py:31)
    size_t pconv_w, pconv_b, pfc_wT, pfc_b, vconv_w, vconv_b, vfc1_wT, vfc1_b, vfc2_w, vfc2_b;
};

// muzero_atari (ref network/py/muzero_atari_network.py:7-70): extra layer groups of the representation and the 601-bin heads
struct DiscreteHeadOffsets { // DiscreteValueNetwork (ref network_unit.py:67-87)
    int hc = 0, hidden = 0, size = 0;
    size_t conv_w = 0, conv_b = 0, fc1_wT = 0, fc1_b = 0, fc2_wT = 0, fc2_b = 0;
};
struct AtariLayers {
    ConvLayer conv1{}, conv2{};            // stride-2 stem convs (96 -> 48 -> 24)
    std::vector<ConvLayer> rb1, rb2, rb3; // one residual block each (2 convs) at 48x48xC/2, 24x24xC, 12x12xC
    std::vector<ConvLayer> tail;          // num_blocks residual blocks at h x w
    DiscreteHeadOffsets reward, value;
};

// host-side weight utilities (weights.cpp)
long netParamCount(const mz_net_desc& d);
bool netGenerate(const mz_net_desc& d, uint64_t seed, float* out);
bool netValidateDesc(const mz_net_desc& d);
bool packWeights(const mz_net_desc& d, const float* raw, size_t n, std::vector<float>& packed, std::vector<ConvLayer>& repr, std::vector<ConvLayer>& dyn,
                 HeadOffsets& h, AtariLayers& at);

bool readTorchScript(const std::string& path, mz_net_desc* desc, std::vector<float>* weights, std::string* err); // ptfile.cpp
// Network::loadModel's file access (ref network/network.h:18-37): `path` as the reference's TorchScript archive; if "x.pt" does not exist,
// the sibling flat blob "x.mzw" (minizero_amd/export_weights.py); validated against the hyper-parameters.  Sets the error string on failure.
bool readWeightFile(const std::string& path, mz_net_desc* desc, std::vector<float>* weights); // capi.cpp

float invertValueHost(float value); // 601-bin decode helper (ref utils/utils.h:102-108)
int invertValuesOnDevice(int device, const float* values, int n, float* out); // the device twin (net_atari_body.h), test access

struct TowerArgs;
struct TowerArgsBf16;
struct HeadParams;
struct AtariHeadParams;
struct PoolView;
struct GoDevView;
struct GumbelView;
class Pool;

// What ONE worker wants of the MuZero simulation kernels (several workers may run on one Net — mz_worker_create_shared — so these are launch arguments,
// not state of the network): four workgroups per game where the pool leaves CUs idle, Gumbel rounds evaluated ahead, the slab's second bank
struct SimMzMode {
    bool cluster = true;  // false: always one workgroup per game
    bool rounds = false;  // the worker evaluates Gumbel rounds ahead (simPreEvalMz): the cluster kernel then runs every game's 601-bin heads alone
    int alt_base = 0;     // != 0: the slab has 2 x alt_base slots per game, the upper half for the rounds' second expected leaves (sim.hip simPreProbe)
};

class Net {
public:
    Net() = default;
    ~Net();
    int init(int device, const mz_net_desc& d, const float* raw, size_t n);
    int reload(const float* raw, size_t n);

    // device-pointer entry points (all on stream_)
    // in_bits: d_feat holds bit-packed planes (GameEnv::featureBits layout) instead of f32 planes
    int forwardAZ(const float* d_feat, int B, float* d_policy, float* d_logit, float* d_value, bool in_bits = false);
    bool hasFusedTower();
    int initial(const float* d_feat, int B, float* d_policy, float* d_logit, float* d_value, float* d_hidden, const int* d_dst_idx);
    // hidden source: d_hidden_src[(src_idx ? src_idx[b] : b)][C][P]; action: planes (d_action_planes) or ids (d_action_ids)
    int recurrent(const float* d_hidden_src, const int* d_src_idx, const float* d_action_planes, const int* d_action_ids, int B, float* d_policy,
                  float* d_logit, float* d_value, float* d_reward, float* d_hidden_dst, const int* d_dst_idx);
    // host/device wrappers used by the C ABI
    int forwardAZ_any(const float* feat, int B, float* policy, float* logit, float* value, int where);
    int initial_any(const float* feat, int B, float* policy, float* logit, float* value, float* hidden, int where);
    int recurrent_any(const float* hidden_in, const float* action, int B, float* policy, float* logit, float* value, float* reward, float* hidden_out,
                      int where);
    // the per-game simulation kernel (sim.hip): `nsims` whole simulations (select, leaf environment, tower, heads, candidates, expand +
    // backup) of every game in ONE launch, each game advancing on its own workgroup.  *launched = false when no instance fits.
    // d_root_noise ([games][A], nullable): Dirichlet noise applied to the root children before simulation 1
    // noise_kind 1: Dirichlet on the priors, 2: Gumbel on the logits; gum != nullptr: Gumbel root logic on the device (d_start: [games] scratch;
    // host_start: the first simulation starts from d_start as uploaded by the host)
    int simLaunch(Pool& pool, const GoDevView& gv, float* d_policy, float* d_logit, float* d_value, const uint8_t* d_rot, int sim0, int nsims,
                  bool* launched, const float* d_root_noise = nullptr, float noise_eps = 0.0f, int noise_kind = 1, const struct GumbelView* gum = nullptr,
                  int* d_start = nullptr, bool host_start = false);
    // num_simulation: the kernels keep per-search tables / the path in LDS; searches too long for 160 KB use the lock-step kernels
    bool hasSimKernel(int board_n, int env_kind = 0, int num_simulation = 0) const; // env_kind: GoDevView::kind
    // ... on the one-tile tower (sim_wide.inc, sim_wide_a.hip): Go with 128 / 256 hidden channels or on 7x7 / 13x13 / 19x19 boards
    bool hasSimKernelWide(int board_n, int env_kind, int num_simulation) const;
    bool simWidePlan(int board_n, int env_kind, int num_simulation, const HeadParams& hp, int channels, int W32, size_t leaf_bytes, size_t scratch_bytes, int* lf, size_t* lds,
                     size_t* tile_bytes_out) const;
    int simLaunchWide(const struct SimArgs& a, const GoDevView& gv, int max_depth, const uint8_t* d_rot, int sim0, int nsims, bool host_start, int lf, size_t lds, bool* launched);
    int uploadSimArgs(const struct SimArgs& a);
    // MuZero (board games): the same for initial + recurrent inference; hidden states live in the caller's slab [games][slots][C * P]
    // nsims simulations (slots sim0 ..) of every game in one launch of sim_kernel_mz; muzero_atari: sim0 >= 1 (the root's 96x96 representation
    // runs as stand-alone kernels), value / reward come out of the kernel in game scale (d_reward: [games]).  root_given (sim0 == 0, nsims == 1): the
    // root's network outputs already lie in d_policy / d_logit / d_value / d_reward (transformed scale) and its hidden state in slab slot 0 — written by
    // initial() — and simulation 0 is only the root's candidate list + expand + backup, one workgroup per game
    int simLaunchMz(Pool& pool, float* d_hidden, int slots, const unsigned* d_root_feat, const unsigned long long* d_root_legal, const int* d_root_turn,
                    int num_players, float* d_policy, float* d_logit, float* d_value, float* d_reward, int sim0, int nsims, bool* launched,
                    const float* d_root_noise, float noise_eps, int noise_kind, const GumbelView* gum, int* d_start, bool host_start, bool root_given = false,
                    int pre_epoch = 0, bool noise_applied = false, const SimMzMode& mode = SimMzMode());
    // Gumbel rounds (sim.hip sim_pre_kernel_mz, muzero_atari): the leaves the next R simulations of every game are expected to reach, evaluated side by
    // side into the entries of simulations s0 .. s0 + R - 1, tagged with `epoch` (a per-move serial number, != 0); the following simLaunchMz calls of the
    // move pass the same pre_epoch and consume the entries that turn out to be their leaves.  simRootNoiseMz applies the root noise as a launch of its own
    // (the first round needs the noisy logits before simulation 1): the simLaunchMz calls after it pass noise_applied = true.
    // want_alt: the round's second expected leaves too (where they fit the chip beside the first); pairs: without them, two workgroups per leaf where those fit
    // (sim.hip sim_pre_pair_kernel_mz)
    int simPreEvalMz(int games, int max_depth, int s0, int R, int epoch, bool* launched, bool want_alt = true, bool pairs = false);
    // the same evaluation as a pipeline of batched kernels (sim_rounds.hip: walks | trunks of several leaves per workgroup | the heads' FC layers as MFMA GEMMs over
    // all leaves | per-leaf tails); the entries it writes are bit-identical.  *launched = false: no instance for this network (use simPreEvalMz)
    int simPreEvalBatchMz(int games, int max_depth, int s0, int R, int epoch, bool* launched, int force_nl = 0, bool want_alt = false); // force_nl: leaves per trunk workgroup (0 = by the pool size)
    bool hasPreBatch() const;
    bool pairsAvailable() const { return pair_ok_; }
    void pairTrouble() { pair_ok_ = false; } // a pair launch met a partner that never showed up (pre_stat[129]): one workgroup per leaf from here on
    int cuCount() const { return cu_count_; }
    int simRootNoiseMz(int games);
    int simPreStats(unsigned* hits, unsigned* evals, unsigned* alt_hits);
    int simPreCountersAsync(unsigned* h_pinned);
    // serial numbers of the moves whose leaves are evaluated ahead come from the NETWORK: the entries (pre_key_ / pre_out_) belong to it, and two workers on one
    // network that both counted from 1 would take each other's stale entries for their own (same parent slot, same action, same number)
    int nextPreEpoch() { pre_epoch_counter_ = pre_epoch_counter_ == 0x7fffffff ? 1 : pre_epoch_counter_ + 1; return pre_epoch_counter_; }
    bool hasSimKernelMz(int num_simulation = 0) const;
    bool hasPreBoard() const; // MuZero board games: is there a kernel that evaluates a Gumbel round's leaves ahead for this shape?
    // MuZero board games on the one-tile tower (sim_wide_mz.hip sim_kernel_mz_wide)
    bool simMzWidePlan(int num_simulation, int* lf, size_t* lds, size_t* tile_bytes_out, int* c0q_out, int* cdq_out) const;
    int simLaunchMzWide(const struct SimArgs& a, int games, int sim0, int nsims, int host_start, int lf, size_t lds, int c0q, int cdq, bool* launched);
    int expandAtariFeatures(const uint8_t* d_raw, int raw_bytes, int B, float* d_feat);
    int shiftExpandAtariFeatures(const uint8_t* d_prev, const uint8_t* d_newest, const uint8_t* d_meta, uint8_t* d_cur, int raw_bytes, int B, float* d_feat); // raw observations -> float planes (net_atari.hip)
    void makeAtariHeadParams(AtariHeadParams* out) const; // net_atari.hip
    // opt-in 16-bit-input tower (net_bf16_body.h): 0 = f32 (default, bit-exact against the oracle), 1 = bf16x3 (split bf16 operands on
    // v_mfma_f32_16x16x32_bf16, f32 accumulation; outputs within 1e-3 of the f32 path, records not bit-identical to the reference)
    int setPrecision(int mode);
    int precision() const { return precision_; }
    bool bf16Supported() const;
    int timeForward(int B, int iters, float* ms_total, float* ms_conv, double* conv_flops);
    int timeTowerConv(int B, int iters, float* ms_per_launch, double* flops_per_launch, double* bytes_per_launch);

    mz_net_desc desc_{};
    int device_ = -1;
    hipStream_t stream_ = nullptr;
    bool own_stream_ = false;
    int P() const { return desc_.hidden_channel_height * desc_.hidden_channel_width; }
    int hiddenSize() const { return desc_.num_hidden_channels * P(); }
    int featSize() const { return desc_.num_input_channels * desc_.input_channel_height * desc_.input_channel_width; }

private:
    int ensureBatch(int B);
    int runTrunk(const std::vector<ConvLayer>& t, const float* d_in, int B, float** d_out, bool in_bits = false);
    int launchConv(const ConvLayer& L, const float* in, const float* skip, float* out, int B);
    int launchTower(const std::vector<ConvLayer>& t, const float* in, float* out, int B, bool* launched, bool in_bits, bool has_stem = true);
    int launchHeads(const float* x, int B, float* policy, float* logit, float* value, float* hidden_dst, const int* dst_idx, bool scale_hidden);
    bool makeTowerArgs(const std::vector<ConvLayer>& t, bool in_bits, bool has_stem, TowerArgs* out, int* c0) const;
    // shapes beyond the two-tile tower's (net_wide.hip): one-tile tower for 128 / 256 channels and 7x7 .. 19x19 boards; a run-time-shaped conv3x3 behind everything
    bool makeWideArgs(const std::vector<ConvLayer>& t, bool in_bits, TowerArgs* out, int* c0q) const;
    bool hasWideTower(const std::vector<ConvLayer>& t) const;
    int launchTowerWide(const std::vector<ConvLayer>& t, const float* in, float* out, float* tmp, int B, bool* launched, bool in_bits);
    int launchConvAny(const ConvLayer& L, const float* in, const float* skip, float* out, int B);
    int unpackBits(const float* d_bits, int C, int B, float* d_feat);
    DevBuf<float> unpacked_;    // f32 planes of a bit-packed batch (run-time-shaped kernels only)
    bool makeTowerArgsBf16(TowerArgsBf16* out) const;
    int packBf16(const std::vector<float>& packed);
    int launchTowerBf16(const float* d_feat, float* out, int B, bool in_bits);
    int precision_ = 0;
    DevBuf<uint4> wfrag_;                 // bf16 hi / lo A-fragments of the representation trunk (built at load when the shape is supported)
    std::vector<unsigned> wfrag_off_;     // per layer, in 16-byte units
    DevBuf<unsigned> bits_in_;            // bf16 path: f32 0/1 planes packed to bits
    void makeHeadParams(HeadParams* out) const;

    DevBuf<float> params_;
    std::vector<ConvLayer> repr_, dyn_;
    HeadOffsets heads_{};
    AtariLayers at_;
    DevBuf<float> at_buf_[3];   // muzero_atari representation activations (largest stage: [B][C/2][H/2][W/2])
    int at_batch_ = 0;
    DevBuf<float> at_out_;      // the representation's 6x6 output of the whole batch (the two halves of initialAtari write their parts)
    static constexpr int kReprParts = 4;
    hipStream_t at_streams_[kReprParts - 1] = {}; // the other parts of the batch: their layers run beside the first part's, a phase apart (initialAtari)
    hipEvent_t at_fork_ = nullptr, at_join_[kReprParts - 1] = {};
    int initialAtari(const float* d_feat, int B, float* d_policy, float* d_logit, float* d_value, float* d_hidden, const int* d_dst_idx);
    int recurrentAtari(const float* d_hidden_src, const int* d_src_idx, const float* d_action_planes, const int* d_action_ids, int B, float* d_policy,
                       float* d_logit, float* d_value, float* d_reward, float* d_hidden_dst, const int* d_dst_idx);
    int max_batch_ = 0;
    DevBuf<float> act_[3];      // [B][C][P] ping-pong + residual temp
    DevBuf<float> rec_in_;      // [B][C + a][P] dynamics input
    DevBuf<float> io_in_, io_in2_, io_policy_, io_logit_, io_value_, io_reward_, io_hidden_; // staging for MZ_HOST callers
    bool conv_only_ = false;    // timing mode: skip the heads
    DevBuf<char> sim_args_;     // SimArgs block of sim_kernel in device memory + the host copy it was uploaded from
    std::vector<char> sim_args_host_;
    DevBuf<unsigned long long> sim_prof_; // MZ_SIM_PROF=1: per-phase tick counters of sim_kernel
    void dumpSimProf();
    void dumpRoundsProf(); // sim_rounds.hip: MZ_SIM_PROF=1, the shader clock under the multi-leaf trunks
    DevBuf<unsigned> sim_sink_;
    DevBuf<char> sim_cluster_mem_; // cluster mode of the MuZero simulation kernel (sim_cluster.h): per-game exchange blocks
    DevBuf<int> pre_key_;          // leaves evaluated ahead: keys [games][slots][4], outputs [policy | logit | value | reward], counters
    DevBuf<float> pre_out_;
    DevBuf<unsigned> pre_stat_;
    DevBuf<char> pre_pair_mem_;    // exchange blocks of the leaves evaluated by pairs of workgroups (sim.hip sim_pre_pair_kernel_mz)
    bool pair_ok_ = true;          // ... cleared when a placement probe or a cooperative launch says no
    int pair_checked_ = 0, pair_lpad_ = 0, pair_set_ = 0, pair_clean_ = 0; // launch shape probed / allocated for, the set the next launch uses, its blocks known to be zero
public:
    unsigned long long pre_pair_launches_ = 0;
private:
    DevBuf<char> pre_ctl_, pre_f_, pre_h1_, pre_lg_; // batched round evaluation (sim_rounds.hip): leaf table, FC1 inputs, hidden units, bins
    int cu_count_ = 0;
    int sim_cluster_checked_ = 0; // pool size (padded) whose cluster placement has been probed
    bool coop_launch_ = false;
public:
    int pre_epoch_counter_ = 0;
    bool sim_octet_ = true;     // cluster mode: the 601-bin heads of the games that share an XCD are computed together (sim_cluster.h octetHead)
    bool sim_cluster_ = true;   // the device can run four workgroups per game (muzero_atari instances; MZ_SIM_CLUSTER=0, a failed placement probe or a refused cooperative launch clear it)
    bool use_fused_ = true;     // fused persistent tower kernel (same arithmetic as the per-layer kernels)
};

} // namespace mz
