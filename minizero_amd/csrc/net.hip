// gfx950 (MI355X / CDNA4) inference kernels of libmzgpu.  Built with -ffp-contract=off: every fused
// multiply-add below is explicit, so the results are a pure function of the arithmetic order written in
// DESIGN.md §"Network numerics" (the parity tests compare against a CPU restatement of that order).
//
// conv3x3_mfma  — im2col-free 3x3 convolution (pad 1) as an implicit GEMM on the f32 MFMA pipe:
//                 D[oc][pixel] += W'[oc][k] * X[k][pixel], k = (tap, channel) tap-major.
//                 One workgroup (4 wave64) per sample; the whole zero-padded input image of the sample
//                 ([C_in][H+2][W+2] f32, <= 39 KB) is staged once in LDS, the 9 taps are 9 shifted LDS
//                 reads of the same tile.  A-fragments (weights) are pre-packed on the host so one
//                 coalesced 256-B global read per wave = one v_mfma_f32_16x16x4_f32 A operand, and they
//                 are double-buffered one tap ahead in registers.  Epilogue fuses folded-BN bias,
//                 residual skip and ReLU and writes 64-B contiguous pixel runs.
// heads_kernel  — policy head (conv1x1+BN+ReLU+FC+softmax) and value head (conv1x1+BN+ReLU+FC+ReLU+FC+tanh)
//                 fused in one workgroup per sample; for MuZero also the per-sample min/max rescale of
//                 the hidden state (ref muzero_network.py:154-164) and its scatter into the HBM slab.
#include "net.h"
#include "net_body.h"
#include "net_bf16_body.h"
#include <cmath>
#include <cstring>

namespace mz {

// ---------------------------------------------------------------------------------------------
// conv3x3 on the f32 MFMA pipe
// ---------------------------------------------------------------------------------------------
template <int H, int W, int CIN_PAD>
__global__ __launch_bounds__(256) void conv3x3_mfma(const float* __restrict__ in, int cin, const float* __restrict__ wp,
                                                    const float* __restrict__ bias, const float* __restrict__ skip, float* __restrict__ out,
                                                    int cout, int OT)
{
    constexpr int P = H * W, PW = W + 2, CS = planeStride(H, W), CG = CIN_PAD / 4, PT = (P + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float xs[]; // [CIN_PAD][CS]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- stage the zero-padded input image of this sample in LDS ----
    const float* src = in + size_t(b) * cin * P;
    for (int idx = tid; idx < CIN_PAD * CS; idx += 256) {
        int c = idx / CS, pos = idx - c * CS;
        int yy = pos / PW, xx = pos - yy * PW;
        float v = 0.0f;
        if (c < cin && yy >= 1 && yy <= H && xx >= 1 && xx <= W) { v = src[c * P + (yy - 1) * W + (xx - 1)]; }
        xs[idx] = v;
    }
    __syncthreads();

    // B-fragment (activations) base offsets: lane l supplies X[k = l>>4][pixel = 16*pt + (l&15)]
    int pixoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int q = 16 * pt + (lane & 15);
        if (q >= P) { q = 0; } // padding column of the last pixel tile: any valid address, result discarded
        pixoff[pt] = (lane >> 4) * CS + (q / W) * PW + (q % W);
    }

    for (int ot = wave; ot < OT; ot += 4) {
        f32x4 acc[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) { acc[pt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
        const float* wl = wp + size_t(ot) * 64 + lane; // + ((t*CG + cg)*OT)*64
        const size_t wstep = size_t(OT) * 64;
        float a_cur[CG], a_nxt[CG];
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) { a_cur[cg] = wl[size_t(cg) * wstep]; }
#pragma unroll 1
        for (int t = 0; t < 9; ++t) { // runtime tap loop: keeps live ranges to one tap (96 MFMAs) + the next tap's weights
            const int tn = t < 8 ? t + 1 : 8; // last iteration re-reads tap 8 (harmless, keeps the loop body uniform)
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) { a_nxt[cg] = wl[(size_t(tn) * CG + cg) * wstep]; }
            const int tapoff = (t / 3) * PW + (t % 3);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    float bv = xs[pixoff[pt] + cg * 4 * CS + tapoff];
                    acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[cg], bv, acc[pt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) { a_cur[cg] = a_nxt[cg]; }
        }
        // ---- epilogue: D layout col = lane&15 (pixel), row = 4*(lane>>4) + r (oc within the tile) ----
        float* dst = out + size_t(b) * cout * P;
        const float* sk = skip ? skip + size_t(b) * cout * P : nullptr;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int q = 16 * pt + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oc = 16 * ot + 4 * (lane >> 4) + r;
                if (q < P && oc < cout) {
                    float v = acc[pt][r] + bias[oc];
                    if (sk) { v = v + sk[oc * P + q]; }
                    dst[oc * P + q] = v > 0.0f ? v : 0.0f;
                }
            }
        }
    }
}

template <int H, int W, int CIN0_PAD, int CPAD>
__global__ __launch_bounds__(512) void tower_fused(const float* __restrict__ in, const float* __restrict__ params, TowerArgs ta,
                                                   float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float tiles[]; // 3 x [CMAX][CS]
    towerBody<H, W, CIN0_PAD, CPAD>(in, params, ta, out, blockIdx.x, threadIdx.x, tiles);
}

// the opt-in bf16x3 tower as a stand-alone launch (net_bf16_body.h): bit-packed planes in, f32 NCHW out
template <int H, int W>
__global__ __launch_bounds__(512) void tower_fused_bf16(const unsigned* __restrict__ in_bits, const uint4* __restrict__ wfrag, const float* __restrict__ params,
                                                        TowerArgsBf16 ta, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char bufs[];
    towerBodyBf16<H, W>(in_bits, wfrag, params, ta, out, blockIdx.x, threadIdx.x, bufs);
}

// f32 planes that hold 0 / 1 (every board-game plane) -> one bit per point, the input format of the fused towers
__global__ __launch_bounds__(256) void pack_bits_kernel(const float* __restrict__ feat, int C, int P, unsigned* __restrict__ bits)
{
    const int b = blockIdx.x, W32 = (P + 31) / 32;
    for (int i = threadIdx.x; i < C * W32; i += 256) {
        const int c = i / W32, w = i - c * W32;
        unsigned v = 0;
        for (int k = 0; k < 32 && w * 32 + k < P; ++k) { v |= (feat[(size_t(b) * C + c) * P + w * 32 + k] != 0.0f ? 1u : 0u) << k; }
        bits[size_t(b) * C * W32 + i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// dynamics input: cat(hidden[src], action plane) on the channel axis (ref muzero_network.py:32)
// action_mode 1: board games, one-hot position plane (all zero for pass; ref go.cpp:310-315)
// ---------------------------------------------------------------------------------------------
__global__ void build_recurrent_input(const float* __restrict__ hidden, const int* __restrict__ src_idx, const float* __restrict__ planes,
                                      const int* __restrict__ action_ids, int C, int AC, int P, float* __restrict__ out)
{
    const int b = blockIdx.x;
    const int s = src_idx ? src_idx[b] : b;
    const float* h = hidden + size_t(s) * C * P;
    float* o = out + size_t(b) * (C + AC) * P;
    for (int i = threadIdx.x; i < C * P; i += blockDim.x) { o[i] = h[i]; }
    for (int i = threadIdx.x; i < AC * P; i += blockDim.x) {
        float v;
        if (planes) { v = planes[size_t(b) * AC * P + i]; }
        else { v = (AC == 1) ? (i == action_ids[b] ? 1.0f : 0.0f) : ((i / P) == action_ids[b] ? 1.0f : 0.0f); }
        o[C * P + i] = v;
    }
}

__global__ __launch_bounds__(256) void heads_kernel(const float* __restrict__ x, HeadParams hp, float* __restrict__ policy,
                                                    float* __restrict__ logit, float* __restrict__ value, float* __restrict__ hidden_dst,
                                                    const int* __restrict__ dst_idx, int scale_hidden)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // (scale_hidden bit 1: the activations do not fit the LDS — headsBody reads (and rescales) them in global memory: dense planes, "row stride" 0)
    if (scale_hidden & 2) { headsBody(x, hp, policy, logit, value, hidden_dst, dst_idx, scale_hidden & 1, blockIdx.x, threadIdx.x, 256, sm, x + size_t(blockIdx.x) * hp.C * hp.P, hp.P, 0); }
    else { headsBody(x, hp, policy, logit, value, hidden_dst, dst_idx, scale_hidden, blockIdx.x, threadIdx.x, 256, sm); }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
Net::~Net()
{
    dumpSimProf();
    dumpRoundsProf();
    for (hipStream_t s : at_streams_) { if (s) { (void)hipStreamDestroy(s); } }
    if (at_fork_) { (void)hipEventDestroy(at_fork_); }
    for (hipEvent_t e : at_join_) { if (e) { (void)hipEventDestroy(e); } }
    if (own_stream_ && stream_) { (void)hipStreamDestroy(stream_); }
}

int Net::init(int device, const mz_net_desc& d, const float* raw, size_t n)
{
    if (!netValidateDesc(d)) { return MZ_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        setError("no such GPU: device %d of %d (libmzgpu has no CPU path; createNetwork(file, -1) is not supported)", device, count);
        return MZ_ERR_DEVICE;
    }
    desc_ = d;
    device_ = device;
    MZ_HIP(hipSetDevice(device));
    if (!stream_) {
        MZ_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_) == hipSuccess) { cu_count_ = v; }
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeCooperativeLaunch, device_) == hipSuccess) { coop_launch_ = v != 0; }
        if (const char* e = getenv("MZ_SIM_CLUSTER")) { sim_cluster_ = atoi(e) != 0; }
        if (const char* e = getenv("MZ_SIM_OCTET")) { sim_octet_ = atoi(e) != 0; }
        own_stream_ = true;
    }
    return reload(raw, n);
}

int Net::reload(const float* raw, size_t n)
{
    std::vector<float> packed;
    if (!packWeights(desc_, raw, n, packed, repr_, dyn_, heads_, at_)) { return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipStreamSynchronize(stream_));
    if (!params_.ensure(packed.size())) { setError("hipMalloc of %zu parameter floats failed", packed.size()); return MZ_ERR_DEVICE; }
    MZ_HIP(hipMemcpy(params_.p, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    return packBf16(packed);
}

// ---- opt-in bf16x3 tower: fragments, launch, switch ----
bool Net::bf16Supported() const
{
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width;
    return desc_.type == 0 && desc_.num_hidden_channels == 64 && !repr_.empty() && repr_.size() >= 3 && repr_.size() <= 48 && (repr_.size() % 2) == 1 &&
           repr_[0].cin <= 32 && ((H == 9 && W == 9) || (H == 8 && W == 8));
}

static uint16_t bf16Rne(float v)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) { return static_cast<uint16_t>(u >> 16); } // inf / nan: truncate
    u += 0x7FFFu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}
static float bf16ToFloat(uint16_t h)
{
    const uint32_t u = static_cast<uint32_t>(h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int Net::packBf16(const std::vector<float>& packed)
{
    wfrag_off_.clear();
    if (!bf16Supported()) { return MZ_OK; }
    // per layer [tap][oc-tile][k-block][hi, lo][lane][8]: lane = 16 * kg + m holds W'[oc = 16 * ot + m][c = 32 * kb + 8 * kg + j][tap]
    std::vector<uint16_t> frag;
    for (const ConvLayer& L : repr_) {
        const int CG = L.cin_pad / 4, OT = L.cout_pad / 16, KB = (L.cin + 31) / 32;
        wfrag_off_.push_back(static_cast<unsigned>(frag.size() / 8));
        for (int t = 0; t < 9; ++t)
            for (int ot = 0; ot < OT; ++ot)
                for (int kb = 0; kb < KB; ++kb)
                    for (int hl = 0; hl < 2; ++hl)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int oc = 16 * ot + (l & 15), c = 32 * kb + 8 * (l >> 4) + j;
                                float w = 0.0f;
                                if (oc < L.cout && c < L.cin) { w = packed[L.w_off + ((size_t(t) * CG + c / 4) * OT + oc / 16) * 64 + 16 * (c % 4) + oc % 16]; }
                                const uint16_t hi = bf16Rne(w);
                                frag.push_back(hl == 0 ? hi : bf16Rne(w - bf16ToFloat(hi)));
                            }
    }
    if (!wfrag_.ensure(frag.size() / 8)) { setError("hipMalloc of the bf16 fragments failed"); return MZ_ERR_DEVICE; }
    MZ_HIP(hipMemcpy(wfrag_.p, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return MZ_OK;
}

bool Net::makeTowerArgsBf16(TowerArgsBf16* out) const
{
    if (!bf16Supported() || wfrag_off_.size() != repr_.size()) { return false; }
    TowerArgsBf16& ta = *out;
    memset(&ta, 0, sizeof(ta));
    ta.nlayers = static_cast<int>(repr_.size());
    ta.cin0 = repr_[0].cin;
    ta.C = desc_.num_hidden_channels;
    ta.OT = repr_[0].cout_pad / 16;
    for (size_t i = 0; i < repr_.size(); ++i) { ta.w_off[i] = wfrag_off_[i]; ta.b_off[i] = static_cast<unsigned>(repr_[i].b_off); }
    return true;
}

int Net::setPrecision(int mode)
{
    if (mode != 0 && mode != 1) { setError("precision %d unknown (0 = f32, 1 = bf16x3)", mode); return MZ_ERR_ARG; }
    if (mode == 1 && !bf16Supported()) {
        setError("mz_nn_precision=bf16x3 is built for AlphaZero networks with 64 hidden channels on 9x9 / 8x8 boards; this network keeps the f32 tower");
        return MZ_ERR_ARG;
    }
    precision_ = mode;
    return MZ_OK;
}

template <int H, int W>
static int launchTowerBf16T(const TowerArgsBf16& ta, const uint4* wfrag, const float* params, const unsigned* bits, float* out, int B, hipStream_t s)
{
    constexpr size_t lds = towerBf16LdsBytes<H, W>(false);
    MZ_LDS_ATTR((tower_fused_bf16<H, W>), lds);
    hipLaunchKernelGGL((tower_fused_bf16<H, W>), dim3(B), dim3(512), lds, s, bits, wfrag, params, ta, out);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Net::launchTowerBf16(const float* d_feat, float* out, int B, bool in_bits)
{
    TowerArgsBf16 ta;
    if (!makeTowerArgsBf16(&ta)) { setError("bf16x3 tower: unsupported network"); return MZ_ERR_STATE; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width;
    const unsigned* bits = reinterpret_cast<const unsigned*>(d_feat);
    if (!in_bits) {
        const int W32 = (P() + 31) / 32;
        if (!bits_in_.ensure(size_t(B) * ta.cin0 * W32)) { setError("hipMalloc of the packed planes failed"); return MZ_ERR_DEVICE; }
        hipLaunchKernelGGL(pack_bits_kernel, dim3(B), dim3(256), 0, stream_, d_feat, ta.cin0, P(), bits_in_.p);
        MZ_HIP(hipGetLastError());
        bits = bits_in_.p;
    }
    if (H == 9 && W == 9) { return launchTowerBf16T<9, 9>(ta, wfrag_.p, params_.p, bits, out, B, stream_); }
    return launchTowerBf16T<8, 8>(ta, wfrag_.p, params_.p, bits, out, B, stream_);
}

int Net::ensureBatch(int B)
{
    if (B <= max_batch_) { return MZ_OK; }
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipStreamSynchronize(stream_));
    const size_t act = size_t(B) * hiddenSize();
    for (auto& a : act_) { if (!a.alloc(act)) { setError("hipMalloc activations failed"); return MZ_ERR_DEVICE; } }
    if (desc_.type >= 1 && !rec_in_.alloc(size_t(B) * (desc_.num_hidden_channels + desc_.num_action_feature_channels) * P())) {
        setError("hipMalloc dynamics input failed");
        return MZ_ERR_DEVICE;
    }
    max_batch_ = B;
    return MZ_OK;
}

template <int H, int W, int CIN_PAD>
static int launchConvT(const ConvLayer& L, const float* params, const float* in, const float* skip, float* out, int B, hipStream_t s)
{
    constexpr size_t lds = size_t(CIN_PAD) * planeStride(H, W) * sizeof(float);
    MZ_LDS_ATTR((conv3x3_mfma<H, W, CIN_PAD>), lds);
    hipLaunchKernelGGL((conv3x3_mfma<H, W, CIN_PAD>), dim3(B), dim3(256), lds, s, in, L.cin, params + L.w_off, params + L.b_off, skip, out, L.cout,
                       L.cout_pad / 16);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Net::launchConv(const ConvLayer& L, const float* in, const float* skip, float* out, int B)
{
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width;
#define MZ_CONV_CASE(h, w, c) \
    if (H == h && W == w && L.cin_pad == c) { return launchConvT<h, w, c>(L, params_.p, in, skip, out, B, stream_); }
    MZ_CONV_CASE(9, 9, 20)  // Go stem (18 planes)
    MZ_CONV_CASE(9, 9, 64)  // Go tower
    MZ_CONV_CASE(9, 9, 68)  // Go MuZero dynamics stem (64 + 1)
    MZ_CONV_CASE(9, 9, 8)   // small test nets
    MZ_CONV_CASE(9, 9, 12)
    MZ_CONV_CASE(9, 9, 16)
    MZ_CONV_CASE(8, 8, 4)   // Othello stem
    MZ_CONV_CASE(8, 8, 64)
    MZ_CONV_CASE(8, 8, 68)
    MZ_CONV_CASE(8, 8, 8)
    MZ_CONV_CASE(8, 8, 12)
    MZ_CONV_CASE(19, 19, 20) // 19x19 Go (per-layer kernels, lock-step modes)
    MZ_CONV_CASE(19, 19, 64)
    MZ_CONV_CASE(19, 19, 68)
    MZ_CONV_CASE(19, 19, 8)
    MZ_CONV_CASE(19, 19, 12)
    MZ_CONV_CASE(6, 6, 4)   // 6x6 Othello
    MZ_CONV_CASE(6, 6, 8)
    MZ_CONV_CASE(6, 6, 64)
    MZ_CONV_CASE(3, 3, 4)   // TicTacToe stem
    MZ_CONV_CASE(3, 3, 16)
    MZ_CONV_CASE(3, 3, 20)
#undef MZ_CONV_CASE
    return launchConvAny(L, in, skip, out, B); // any other shape: the run-time-shaped kernel (net_wide.hip)
}

template <int H, int W, int CIN0_PAD, int CPAD>
static int launchTowerT(const TowerArgs& ta, const float* params, const float* in, float* out, int B, hipStream_t s)
{
    constexpr int CMAX = CIN0_PAD > CPAD ? CIN0_PAD : CPAD;
    constexpr size_t lds = size_t(kTowerTiles) * CMAX * planeStride(H, W) * sizeof(float);
    MZ_LDS_ATTR((tower_fused<H, W, CIN0_PAD, CPAD>), lds);
    hipLaunchKernelGGL((tower_fused<H, W, CIN0_PAD, CPAD>), dim3(B), dim3(512), lds, s, in, params, ta, out);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

// TowerArgs of a trunk that has the shape the fused kernel handles (false otherwise); *c0 = the CIN0_PAD template argument
bool Net::makeTowerArgs(const std::vector<ConvLayer>& t, bool in_bits, bool has_stem, TowerArgs* out, int* c0) const
{
    if (!use_fused_ || t.size() > 48 || t.empty() || (t.size() % 2) == (has_stem ? 0u : 1u)) { return false; }
    const int C = desc_.num_hidden_channels;
    for (size_t i = has_stem ? 1 : 0; i < t.size(); ++i) { if (t[i].cin != C || t[i].cout != C) { return false; } }
    if (t[0].cout != C || C % 4 != 0) { return false; }
    TowerArgs& ta = *out;
    ta.nlayers = static_cast<int>(t.size());
    ta.cin0 = t[0].cin;
    ta.C = C;
    ta.OT = t[0].cout_pad / 16;
    ta.in_bits = in_bits ? 1 : 0;
    ta.has_stem = has_stem ? 1 : 0;
    for (size_t i = 0; i < t.size(); ++i) { ta.w_off[i] = static_cast<unsigned>(t[i].w4_off); ta.b_off[i] = static_cast<unsigned>(t[i].b_off); }
    *c0 = has_stem ? t[0].cin_pad : C; // without a stem the template's CIN0_PAD is unused: pick the C instance
    return true;
}

// returns MZ_OK and sets *launched when a fused instance exists for this trunk
int Net::launchTower(const std::vector<ConvLayer>& t, const float* in, float* out, int B, bool* launched, bool in_bits, bool has_stem)
{
    *launched = false;
    TowerArgs ta;
    int c0 = 0;
    if (!makeTowerArgs(t, in_bits, has_stem, &ta, &c0)) { return MZ_OK; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
#define MZ_TOWER_CASE(h, w, cin0, cpad) \
    if (H == h && W == w && c0 == cin0 && C == cpad) { *launched = true; return launchTowerT<h, w, cin0, cpad>(ta, params_.p, in, out, B, stream_); }
    MZ_TOWER_CASE(9, 9, 20, 64)  // Go AlphaZero / MuZero representation
    MZ_TOWER_CASE(9, 9, 68, 64)  // Go MuZero dynamics
    MZ_TOWER_CASE(8, 8, 4, 64)   // Othello
    MZ_TOWER_CASE(8, 8, 68, 64)
    MZ_TOWER_CASE(9, 9, 20, 8)   // small test nets
    MZ_TOWER_CASE(9, 9, 12, 8)
    MZ_TOWER_CASE(8, 8, 4, 8)
    MZ_TOWER_CASE(8, 8, 12, 8)
    MZ_TOWER_CASE(3, 3, 4, 16)   // TicTacToe
    MZ_TOWER_CASE(3, 3, 20, 16)
    MZ_TOWER_CASE(6, 6, 84, 64)  // Atari dynamics (64 + 18 action planes)
    MZ_TOWER_CASE(6, 6, 64, 64)  // Atari representation tail (no stem)
    MZ_TOWER_CASE(6, 6, 52, 32)  // small Atari test nets (C = 32)
    MZ_TOWER_CASE(6, 6, 32, 32)
#undef MZ_TOWER_CASE
    return MZ_OK;
}

int Net::runTrunk(const std::vector<ConvLayer>& t, const float* d_in, int B, float** d_out, bool in_bits)
{
    if (precision_ == 1 && &t == &repr_) {
        int rc = launchTowerBf16(d_in, act_[0].p, B, in_bits);
        if (rc) { return rc; }
        *d_out = act_[0].p;
        return MZ_OK;
    }
    bool launched = false;
    int frc = launchTower(t, d_in, act_[0].p, B, &launched, in_bits);
    if (frc) { return frc; }
    if (launched) { *d_out = act_[0].p; return MZ_OK; }
    if ((frc = launchTowerWide(t, d_in, act_[0].p, act_[1].p, B, &launched, in_bits))) { return frc; } // one-tile tower: wide / large-board shapes
    if (launched) { *d_out = act_[0].p; return MZ_OK; }
    if (in_bits) { // the per-layer kernels read f32 planes
        if (!unpacked_.ensure(size_t(B) * t[0].cin * P())) { setError("hipMalloc of the unpacked planes failed"); return MZ_ERR_DEVICE; }
        if ((frc = unpackBits(d_in, t[0].cin, B, unpacked_.p))) { return frc; }
        d_in = unpacked_.p;
    }
    float *x = act_[0].p, *tmp = act_[1].p, *y = act_[2].p;
    int rc = launchConv(t[0], d_in, nullptr, x, B);
    if (rc) { return rc; }
    for (size_t i = 1; i + 1 < t.size(); i += 2) { // ref network_unit.py:14-23
        if ((rc = launchConv(t[i], x, nullptr, tmp, B))) { return rc; }
        if ((rc = launchConv(t[i + 1], tmp, x, y, B))) { return rc; }
        float* s = x; x = y; y = s;
    }
    *d_out = x;
    return MZ_OK;
}

void Net::makeHeadParams(HeadParams* out) const
{
    HeadParams& hp = *out;
    const float* p = params_.p;
    hp.pconv_w = p + heads_.pconv_w; hp.pconv_b = p + heads_.pconv_b; hp.pfc_wT = p + heads_.pfc_wT; hp.pfc_b = p + heads_.pfc_b;
    hp.vconv_w = p + heads_.vconv_w; hp.vconv_b = p + heads_.vconv_b; hp.vfc1_wT = p + heads_.vfc1_wT; hp.vfc1_b = p + heads_.vfc1_b;
    hp.vfc2_w = p + heads_.vfc2_w; hp.vfc2_b = p + heads_.vfc2_b;
    hp.C = desc_.num_hidden_channels; hp.P = P(); hp.A = desc_.action_size; hp.PC = heads_.pc; hp.VH = desc_.num_value_hidden_channels;
}

int Net::launchHeads(const float* x, int B, float* policy, float* logit, float* value, float* hidden_dst, const int* dst_idx, bool scale_hidden)
{
    HeadParams hp;
    makeHeadParams(&hp);
    size_t lds = (size_t(hp.C) * hp.P + size_t(hp.PC) * hp.P + hp.P + hp.VH + hp.A + 16) * sizeof(float);
    int x_global = 0;
    if (lds > 160 * 1024) { // (e.g. 19x19 x 128 channels) the activations stay where they are: the chains read them from global memory
        lds -= size_t(hp.C) * hp.P * sizeof(float);
        x_global = 1;
        if (lds > 160 * 1024) { setError("heads: %zu bytes of LDS needed for the policy / value planes", lds); return MZ_ERR_ARG; }
    }
    if (lds > 48 * 1024) { MZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(heads_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds))); }
    hipLaunchKernelGGL(heads_kernel, dim3(B), dim3(256), lds, stream_, x, hp, policy, logit, value, hidden_dst, dst_idx, (scale_hidden ? 1 : 0) | (x_global ? 2 : 0));
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

// ref utils/utils.h:102-108: inverse of h(x) = sign(x)(sqrt(|x|+1)-1) + eps x; inner part in double, powf in float
float invertValueHost(float value)
{
    const float epsilon = 0.001;
    const float sign_value = (value > 0.0f ? 1.0f : (value == 0.0f ? 0.0f : -1.0f));
    return sign_value * (powf((::sqrt(1 + 4 * epsilon * (::fabs(static_cast<double>(value)) + 1 + epsilon)) - 1) / (2 * epsilon), 2.0f) - 1);
}

bool Net::hasFusedTower()
{
    if (repr_.empty()) { return false; }
    // probe with a zero-sized question: does launchTower have an instance for the representation trunk?
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels, c0 = repr_[0].cin_pad;
    static const int inst[][4] = {{9, 9, 20, 64}, {9, 9, 68, 64}, {8, 8, 4, 64}, {8, 8, 68, 64}, {9, 9, 20, 8}, {9, 9, 12, 8}, {8, 8, 4, 8}, {8, 8, 12, 8},
                                  {3, 3, 4, 16}, {3, 3, 20, 16}};
    if (!use_fused_ || (repr_.size() % 2) == 0) { return false; }
    for (auto& i : inst) { if (i[0] == H && i[1] == W && i[2] == c0 && i[3] == C) { return true; } }
    return hasWideTower(repr_);
}

int Net::forwardAZ(const float* d_feat, int B, float* d_policy, float* d_logit, float* d_value, bool in_bits)
{
    if (desc_.type != 0) { setError("forward() called on a %s network", desc_.type == 1 ? "muzero" : "muzero_atari"); return MZ_ERR_STATE; }
    int rc = ensureBatch(B);
    if (rc) { return rc; }
    float* x = nullptr;
    if ((rc = runTrunk(repr_, d_feat, B, &x, in_bits))) { return rc; }
    if (conv_only_) { return MZ_OK; }
    return launchHeads(x, B, d_policy, d_logit, d_value, nullptr, nullptr, false);
}

int Net::initial(const float* d_feat, int B, float* d_policy, float* d_logit, float* d_value, float* d_hidden, const int* d_dst_idx)
{
    if (desc_.type == 2) { return initialAtari(d_feat, B, d_policy, d_logit, d_value, d_hidden, d_dst_idx); }
    if (desc_.type != 1) { setError("initialInference() called on a non-muzero network"); return MZ_ERR_STATE; }
    int rc = ensureBatch(B);
    if (rc) { return rc; }
    float* x = nullptr;
    if ((rc = runTrunk(repr_, d_feat, B, &x))) { return rc; }
    if (conv_only_) { return MZ_OK; }
    return launchHeads(x, B, d_policy, d_logit, d_value, d_hidden, d_dst_idx, true);
}

int Net::recurrent(const float* d_hidden_src, const int* d_src_idx, const float* d_action_planes, const int* d_action_ids, int B, float* d_policy,
                   float* d_logit, float* d_value, float* d_reward, float* d_hidden_dst, const int* d_dst_idx)
{
    if (desc_.type == 2) {
        return recurrentAtari(d_hidden_src, d_src_idx, d_action_planes, d_action_ids, B, d_policy, d_logit, d_value, d_reward, d_hidden_dst, d_dst_idx);
    }
    if (desc_.type != 1) { setError("recurrentInference() called on a non-muzero network"); return MZ_ERR_STATE; }
    int rc = ensureBatch(B);
    if (rc) { return rc; }
    hipLaunchKernelGGL(build_recurrent_input, dim3(B), dim3(256), 0, stream_, d_hidden_src, d_src_idx, d_action_planes, d_action_ids,
                       desc_.num_hidden_channels, desc_.num_action_feature_channels, P(), rec_in_.p);
    MZ_HIP(hipGetLastError());
    float* x = nullptr;
    if ((rc = runTrunk(dyn_, rec_in_.p, B, &x))) { return rc; }
    if (conv_only_) { return MZ_OK; }
    if (d_reward) { MZ_HIP(hipMemsetAsync(d_reward, 0, size_t(B) * sizeof(float), stream_)); } // board games: no reward head (ref muzero_network.h:129)
    return launchHeads(x, B, d_policy, d_logit, d_value, d_hidden_dst, d_dst_idx, true);
}

// ---- MZ_HOST / MZ_DEVICE wrappers ----
#define MZ_ENSURE(buf, n)                                                          \
    if (!(buf).ensure(n)) { setError("hipMalloc of staging buffer failed"); return MZ_ERR_DEVICE; }

int Net::forwardAZ_any(const float* feat, int B, float* policy, float* logit, float* value, int where)
{
    if (B <= 0 || !feat || !policy || !logit || !value) { setError("forward: bad arguments"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    if (where == MZ_DEVICE) {
        int rc = forwardAZ(feat, B, policy, logit, value);
        if (rc) { return rc; }
        MZ_HIP(hipStreamSynchronize(stream_));
        return MZ_OK;
    }
    const size_t A = desc_.action_size;
    MZ_ENSURE(io_in_, size_t(B) * featSize()); MZ_ENSURE(io_policy_, B * A); MZ_ENSURE(io_logit_, B * A); MZ_ENSURE(io_value_, B);
    MZ_HIP(hipMemcpyAsync(io_in_.p, feat, size_t(B) * featSize() * sizeof(float), hipMemcpyHostToDevice, stream_));
    int rc = forwardAZ(io_in_.p, B, io_policy_.p, io_logit_.p, io_value_.p);
    if (rc) { return rc; }
    MZ_HIP(hipMemcpyAsync(policy, io_policy_.p, B * A * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipMemcpyAsync(logit, io_logit_.p, B * A * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipMemcpyAsync(value, io_value_.p, size_t(B) * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipStreamSynchronize(stream_));
    return MZ_OK;
}

int Net::initial_any(const float* feat, int B, float* policy, float* logit, float* value, float* hidden, int where)
{
    if (B <= 0 || !feat || !policy || !logit || !value || !hidden) { setError("initialInference: bad arguments"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    if (where == MZ_DEVICE) {
        int rc = initial(feat, B, policy, logit, value, hidden, nullptr);
        if (rc) { return rc; }
        MZ_HIP(hipStreamSynchronize(stream_));
        return MZ_OK;
    }
    const size_t A = desc_.action_size, HS = hiddenSize();
    MZ_ENSURE(io_in_, size_t(B) * featSize()); MZ_ENSURE(io_policy_, B * A); MZ_ENSURE(io_logit_, B * A); MZ_ENSURE(io_value_, B);
    MZ_ENSURE(io_hidden_, B * HS);
    MZ_HIP(hipMemcpyAsync(io_in_.p, feat, size_t(B) * featSize() * sizeof(float), hipMemcpyHostToDevice, stream_));
    int rc = initial(io_in_.p, B, io_policy_.p, io_logit_.p, io_value_.p, io_hidden_.p, nullptr);
    if (rc) { return rc; }
    MZ_HIP(hipMemcpyAsync(policy, io_policy_.p, B * A * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipMemcpyAsync(logit, io_logit_.p, B * A * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipMemcpyAsync(value, io_value_.p, size_t(B) * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipMemcpyAsync(hidden, io_hidden_.p, B * HS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipStreamSynchronize(stream_));
    if (desc_.type == 2) { for (int i = 0; i < B; ++i) { value[i] = invertValueHost(value[i]); } } // 601-bin decode (ref muzero_network.h:157-163)
    return MZ_OK;
}

int Net::recurrent_any(const float* hidden_in, const float* action, int B, float* policy, float* logit, float* value, float* reward, float* hidden_out,
                       int where)
{
    if (B <= 0 || !hidden_in || !action || !policy || !logit || !value || !hidden_out) { setError("recurrentInference: bad arguments"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    if (where == MZ_DEVICE) {
        int rc = recurrent(hidden_in, nullptr, action, nullptr, B, policy, logit, value, reward, hidden_out, nullptr);
        if (rc) { return rc; }
        MZ_HIP(hipStreamSynchronize(stream_));
        return MZ_OK;
    }
    const size_t A = desc_.action_size, HS = hiddenSize(), AS = size_t(desc_.num_action_feature_channels) * P();
    MZ_ENSURE(io_in_, B * HS); MZ_ENSURE(io_in2_, B * AS); MZ_ENSURE(io_policy_, B * A); MZ_ENSURE(io_logit_, B * A); MZ_ENSURE(io_value_, B);
    MZ_ENSURE(io_reward_, B); MZ_ENSURE(io_hidden_, B * HS);
    MZ_HIP(hipMemcpyAsync(io_in_.p, hidden_in, B * HS * sizeof(float), hipMemcpyHostToDevice, stream_));
    MZ_HIP(hipMemcpyAsync(io_in2_.p, action, B * AS * sizeof(float), hipMemcpyHostToDevice, stream_));
    int rc = recurrent(io_in_.p, nullptr, io_in2_.p, nullptr, B, io_policy_.p, io_logit_.p, io_value_.p, io_reward_.p, io_hidden_.p, nullptr);
    if (rc) { return rc; }
    MZ_HIP(hipMemcpyAsync(policy, io_policy_.p, B * A * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipMemcpyAsync(logit, io_logit_.p, B * A * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipMemcpyAsync(value, io_value_.p, size_t(B) * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (reward) { MZ_HIP(hipMemcpyAsync(reward, io_reward_.p, size_t(B) * sizeof(float), hipMemcpyDeviceToHost, stream_)); }
    MZ_HIP(hipMemcpyAsync(hidden_out, io_hidden_.p, B * HS * sizeof(float), hipMemcpyDeviceToHost, stream_));
    MZ_HIP(hipStreamSynchronize(stream_));
    if (desc_.type == 2) {
        for (int i = 0; i < B; ++i) {
            value[i] = invertValueHost(value[i]);
            if (reward) { reward[i] = invertValueHost(reward[i]); }
        }
    }
    return MZ_OK;
}

// HIP-event timing on the network's own stream (bench.py roofline leg).  conv FLOPs = 2*MAC of the 3x3
// convolutions of one forward (SURVEY.md §8d counts conv+linear; the 3x3 convs are > 99.8 % of it).
int Net::timeForward(int B, int iters, float* ms_total, float* ms_conv, double* conv_flops)
{
    if (B <= 0 || iters <= 0) { setError("timeForward: bad arguments"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    const size_t A = desc_.action_size;
    const bool mz = desc_.type == 1;
    MZ_ENSURE(io_in_, size_t(B) * (mz ? hiddenSize() : featSize())); MZ_ENSURE(io_policy_, B * A); MZ_ENSURE(io_logit_, B * A); MZ_ENSURE(io_value_, B);
    MZ_ENSURE(io_hidden_, size_t(B) * hiddenSize()); MZ_ENSURE(io_in2_, size_t(B) * desc_.num_action_feature_channels * P()); MZ_ENSURE(io_reward_, B);
    std::vector<float> h(io_in_.n);
    uint64_t s = 0x1234567ULL;
    for (auto& v : h) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; v = ((s >> 33) & 1) ? 1.0f : ((s >> 34) & 3) == 0 ? 0.5f : 0.0f; }
    MZ_HIP(hipMemcpy(io_in_.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    MZ_HIP(hipMemset(io_in2_.p, 0, io_in2_.n * sizeof(float)));
    auto run = [&]() -> int {
        if (!mz) { return forwardAZ(io_in_.p, B, io_policy_.p, io_logit_.p, io_value_.p); }
        return recurrent(io_in_.p, nullptr, io_in2_.p, nullptr, B, io_policy_.p, io_logit_.p, io_value_.p, io_reward_.p, io_hidden_.p, nullptr);
    };
    hipEvent_t e0, e1;
    MZ_HIP(hipEventCreate(&e0));
    MZ_HIP(hipEventCreate(&e1));
    float out[2] = {0, 0};
    for (int mode = 0; mode < 2; ++mode) {
        conv_only_ = (mode == 1);
        int rc = run(); // warm-up
        if (rc) { conv_only_ = false; return rc; }
        MZ_HIP(hipStreamSynchronize(stream_));
        MZ_HIP(hipEventRecord(e0, stream_));
        for (int i = 0; i < iters; ++i) { if ((rc = run())) { conv_only_ = false; return rc; } }
        MZ_HIP(hipEventRecord(e1, stream_));
        MZ_HIP(hipEventSynchronize(e1));
        MZ_HIP(hipEventElapsedTime(&out[mode], e0, e1));
        out[mode] /= iters;
    }
    conv_only_ = false;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const std::vector<ConvLayer>& t = mz ? dyn_ : repr_;
    double fl = 0;
    for (auto& L : t) { fl += 2.0 * 9.0 * L.cin * L.cout * P(); }
    if (ms_total) { *ms_total = out[0]; }
    if (ms_conv) { *ms_conv = out[1]; }
    if (conv_flops) { *conv_flops = fl * B; }
    return MZ_OK;
}

// Average duration of ONE launch of the dominant kernel (the residual-tower conv3x3, C -> C with fused bias+ReLU)
// from HIP events on this network's stream, with its algorithmic FLOPs and compulsory HBM bytes per launch.
int Net::timeTowerConv(int B, int iters, float* ms_per_launch, double* flops_per_launch, double* bytes_per_launch)
{
    if (B <= 0 || iters <= 0 || repr_.size() < 2) { setError("timeTowerConv: bad arguments or network without residual blocks"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device_));
    int rc = ensureBatch(B);
    if (rc) { return rc; }
    const ConvLayer& L = repr_[1];
    std::vector<float> h(size_t(B) * hiddenSize());
    uint64_t s = 0x9876543ULL;
    for (auto& v : h) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; v = static_cast<float>((s >> 40) & 0xFFFF) / 65536.0f - 0.25f; }
    MZ_HIP(hipMemcpy(act_[0].p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) { if ((rc = launchConv(L, act_[0].p, nullptr, act_[1].p, B))) { return rc; } }
    hipEvent_t e0, e1;
    MZ_HIP(hipEventCreate(&e0));
    MZ_HIP(hipEventCreate(&e1));
    MZ_HIP(hipStreamSynchronize(stream_));
    MZ_HIP(hipEventRecord(e0, stream_));
    for (int i = 0; i < iters; ++i) { if ((rc = launchConv(L, act_[0].p, nullptr, act_[1].p, B))) { return rc; } }
    MZ_HIP(hipEventRecord(e1, stream_));
    MZ_HIP(hipEventSynchronize(e1));
    float ms = 0;
    MZ_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ms_per_launch) { *ms_per_launch = ms / iters; }
    if (flops_per_launch) { *flops_per_launch = 2.0 * 9.0 * L.cin * L.cout * P() * B; }
    // compulsory traffic: read the input activations once, write the outputs once, read the layer's weights once
    if (bytes_per_launch) { *bytes_per_launch = 4.0 * (double(B) * L.cin * P() + double(B) * L.cout * P() + 9.0 * L.cin * L.cout + L.cout); }
    return MZ_OK;
}

} // namespace mz
