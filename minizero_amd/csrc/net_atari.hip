// muzero_atari inference path (ref network/py/muzero_atari_network.py:7-198, network_unit.py:67-87): the
// 96x96 representation stem (two stride-2 convs, three residual blocks at 48/24/12, two 3x3 average pools),
// then the 6x6 towers (fused, net.hip) and the 601-bin value / reward heads.
//
// conv3x3_tiled   — tiled variant of the MFMA conv for planes that do not fit LDS whole: one workgroup computes an
//                   8 x 16 output tile for every output channel; the (8-1)*S+3 x (16-1)*S+3 input patch of all input
//                   channels is staged in LDS (46-72 KB), a pixel tile of the MFMA is one output row of 16 columns, so
//                   the B-fragment is a stride-S LDS read.  Same tap-major / channel-ascending fmaf chain as everywhere.
// avgpool3s2      — AvgPool2d(3, stride 2, padding 1), count_include_pad: (ky,kx)-ordered sum / 9.
// heads_atari     — reward head (on the UN-scaled hidden state), min/max rescale + slab scatter, policy head, value head;
//                   the 601-way softmax expectation is computed in index order; invertValue stays on the host (double libm).
#include "net.h"
#include <cstdlib>
#include "net_dev.h"
#include "net_body.h"
#include "net_atari_body.h"

namespace mz {

// OT = oc-tiles of the layer, OTW = oc-tiles one workgroup computes (blockIdx.z takes the others): the late layers of the representation have few pixel tiles
// (24x24: 6 per sample, 12x12: 2), and with all output channels in one workgroup a batch of 64 samples is 128-384 workgroups of 1152 dependent-rate MFMAs per
// wave on 256 CUs — split by output channels they are 512-1536 workgroups of 288 (the patch is staged once per workgroup: its loads come from the L2)
template <int STRIDE, int CIN_PAD, int OT, int OTW = OT>
__global__ __launch_bounds__(256) void conv3x3_tiled(const float* __restrict__ in, int cin, int H, int W, const float* __restrict__ wp,
                                                     const float* __restrict__ bias, const float* __restrict__ skip, float* __restrict__ out, int cout)
{
    constexpr int TH = 8, TW = 16, PR = (TH - 1) * STRIDE + 3, PC = (TW - 1) * STRIDE + 3, PLANE = PR * PC, CG = CIN_PAD / 4;
    constexpr int NGROUPS = 4 / OTW, ROWS = TH / NGROUPS;
    static_assert(OT % OTW == 0 && 4 % OTW == 0, "oc-tiles per workgroup");
    extern __shared__ __attribute__((aligned(16))) float xs[]; // [CIN_PAD][PR][PC]
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1, tiles_x = (Wo + TW - 1) / TW;
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    const float* src = in + size_t(b) * cin * H * W;
    // The patch: a thread keeps its POSITIONS of the patch plane (row, column, the bounds check and the source offset are computed once per position) and
    // walks the channels, eight loads in flight at a time.  (One flat index per element — two divisions by PLANE and PC each — was more vector work than
    // the layer's MFMAs: conv1 of the 96x96 representation 73 -> 5x us per 64 samples; before that, one load at a time, 120 us.)
    constexpr int U = CIN_PAD < 32 ? CIN_PAD : 32, NPOS = (PLANE + 255) / 256; // loads in flight per thread: a batch of U is one trip to the L2 / HBM
    static_assert(CIN_PAD % U == 0, "channels are staged U at a time");
    const size_t HW = size_t(H) * W;
#pragma unroll
    for (int k = 0; k < NPOS; ++k) {
        const int pos = k * 256 + tid;
        const int r = pos / PC, q = pos - r * PC;
        const int iy = oy0 * STRIDE + r - 1, ix = ox0 * STRIDE + q - 1;
        const bool inside = pos < PLANE && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float* p = src + (inside ? size_t(iy) * W + ix : 0);
#pragma unroll 1
        for (int c0 = 0; c0 < CIN_PAD; c0 += U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = inside && c0 + u < cin;
                const float x = p[ok ? size_t(c0 + u) * HW : 0];
                v[u] = ok ? x : 0.0f;
            }
            if (pos < PLANE) {
#pragma unroll
                for (int u = 0; u < U; ++u) { xs[(c0 + u) * PLANE + pos] = v[u]; }
            }
        }
    }
    __syncthreads();
    const int ot = static_cast<int>(blockIdx.z) * OTW + wave % OTW, row0 = (wave / OTW) * ROWS;
    f32x4 acc[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) { acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const int lane_off = (lane >> 4) * PLANE + (lane & 15) * STRIDE;
    const float* wl = wp + size_t(ot) * 64 + lane;
    constexpr size_t wstep = size_t(OT) * 64;
    float a_cur[CG], a_nxt[CG];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) { a_cur[cg] = wl[size_t(cg) * wstep]; }
    // B operand (one ds_read_b32 per MFMA): the values of k-group (t, cg + 1) are read BEFORE the MFMAs of group (t, cg) are issued, pinned with
    // sched_barrier — left alone the scheduler puts every read in front of its MFMA and a wave issues one group per LDS round trip (net_body.h tower_layer)
    float bc[ROWS];
    auto bload = [&](float (&bv)[ROWS], int tapoff, int cg) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { bv[j] = xs[lane_off + cg * 4 * PLANE + (row0 + j) * STRIDE * PC + tapoff]; }
    };
    bload(bc, 0, 0);
    // the epilogue's operands (bias, residual input) are fetched now: asked for behind the MFMA loop, every workgroup of a CU — they run in step — sat through
    // the round trip with nothing else to issue (tools/ab_atari_root.sh with the phases switched off one by one: staging 10, MFMAs 28, epilogue 11 of 48 us)
    float* dst = out + size_t(b) * cout * Ho * Wo;
    const float* sk = skip ? skip + size_t(b) * cout * Ho * Wo : nullptr;
    const int ox = ox0 + (lane & 15);
    float ebias[4], eskip[ROWS][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oc = 16 * ot + 4 * (lane >> 4) + r;
        ebias[r] = bias[oc < cout ? oc : 0];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int oy = oy0 + row0 + j;
            const bool ok = sk && oy < Ho && ox < Wo && oc < cout;
            const float x = (sk ? sk : bias)[ok ? (size_t(oc) * Ho + oy) * Wo + ox : 0];
            eskip[j][r] = ok ? x : 0.0f;
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const int tn = t < 8 ? t + 1 : 8;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) { a_nxt[cg] = wl[(size_t(tn) * CG + cg) * wstep]; }
        const int tapoff = (t / 3) * PC + (t % 3), tapoff_n = (tn / 3) * PC + (tn % 3);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            float bn[ROWS];
            if (cg + 1 < CG) { bload(bn, tapoff, cg + 1); } else { bload(bn, tapoff_n, 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[cg], bc[j], acc[j], 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { bc[j] = bn[j]; }
        }
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) { a_cur[cg] = a_nxt[cg]; }
    }
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int oy = oy0 + row0 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = 16 * ot + 4 * (lane >> 4) + r;
            if (oy < Ho && ox < Wo && oc < cout) {
                float v = acc[j][r] + ebias[r];
                if (sk) { v = v + eskip[j][r]; }
                dst[(size_t(oc) * Ho + oy) * Wo + ox] = v > 0.0f ? v : 0.0f;
            }
        }
    }
}

__global__ void avgpool3s2_kernel(const float* __restrict__ in, int C, int H, int W, float* __restrict__ out, int total)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x = i % Wo, y = (i / Wo) % Ho;
        const int bc = i / (Wo * Ho); // sample * C + channel
        const float* p = in + size_t(bc) * H * W;
        float acc = 0.0f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = 2 * y + ky - 1, xx = 2 * x + kx - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) { acc = acc + p[yy * W + xx]; }
            }
        out[i] = acc / 9.0f;
    }
}

__global__ __launch_bounds__(1024) void heads_atari_kernel(const float* __restrict__ x, AtariHeadParams hp, float* __restrict__ policy,
                                                           float* __restrict__ logit, float* __restrict__ value, float* __restrict__ reward,
                                                           float* __restrict__ hidden_dst, const int* __restrict__ dst_idx, int do_reward)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x;
    float* hd = hidden_dst + size_t(dst_idx ? dst_idx[b] : b) * hp.C * hp.P;
    atariHeadsBody<512>(x + size_t(b) * hp.C * hp.P, nullptr, 0, 0, hp, policy, logit, value, reward, hd, do_reward, 0, b, threadIdx.x, sm);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int STRIDE, int CIN_PAD, int OT>
static int launchTiledT(const ConvLayer& L, const float* params, const float* in, const float* skip, float* out, int B, int H, int W, hipStream_t s, int cus)
{
    constexpr int PR = 7 * STRIDE + 3, PC = 15 * STRIDE + 3;
    constexpr size_t lds = size_t(CIN_PAD) * PR * PC * sizeof(float);
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const int tiles = ((Wo + 15) / 16) * ((Ho + 7) / 8);
    // fewer than two workgroups per CU: one oc-tile per workgroup (blockIdx.z = the oc-tile).  Measured at batch 64: 12x12 37 -> 19 us and 41 -> 20 us, 24x24 52 -> 49
    // and 64 -> 55 us; the stride-2 layer 48 -> 24 with its 72-KB patch per workgroup got slower (37 -> 48 us) and keeps all its oc-tiles together
    if (OT > 1 && STRIDE == 1 && tiles * B < 2 * cus) {
        MZ_LDS_ATTR((conv3x3_tiled<STRIDE, CIN_PAD, OT, 1>), lds);
        hipLaunchKernelGGL((conv3x3_tiled<STRIDE, CIN_PAD, OT, 1>), dim3(tiles, B, OT), dim3(256), lds, s, in, L.cin, H, W, params + L.w_off, params + L.b_off, skip, out, L.cout);
        MZ_HIP(hipGetLastError());
        return MZ_OK;
    }
    MZ_LDS_ATTR((conv3x3_tiled<STRIDE, CIN_PAD, OT>), lds);
    hipLaunchKernelGGL((conv3x3_tiled<STRIDE, CIN_PAD, OT>), dim3(tiles, B), dim3(256), lds, s, in, L.cin, H, W, params + L.w_off, params + L.b_off, skip, out, L.cout);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int launchConvAnyStrided(const ConvLayer& L, int stride, const float* params, const float* in, const float* skip, float* out, int B, int H, int W, hipStream_t s); // net_wide.hip
int launchConvBand(const ConvLayer& L, const float* params, const float* in, const float* skip, float* out, int B, int H, int W, hipStream_t s, int cus, bool* launched); // net_wide.hip

static int launchTiled(const ConvLayer& L, int stride, const float* params, const float* in, const float* skip, float* out, int B, int H, int W,
                       hipStream_t s, int cus)
{
    const int ot = L.cout_pad / 16;
    static const bool band = getenv("MZ_ATARI_BAND") != nullptr; // (experiment: the stride-1 layers of the representation on net_wide.hip conv3x3_band)
    if (band && stride == 1) {
        bool launched = false;
        const int rc = launchConvBand(L, params, in, skip, out, B, H, W, s, cus, &launched);
        if (rc != MZ_OK || launched) { return rc; }
    }
#define MZ_TILED_CASE(st, c, o) \
    if (stride == st && L.cin_pad == c && ot == o) { return launchTiledT<st, c, o>(L, params, in, skip, out, B, H, W, s, cus); }
    MZ_TILED_CASE(2, 32, 2) // conv1 32 -> 32 (C = 64)
    MZ_TILED_CASE(1, 32, 2) // residual block at C/2 = 32
    MZ_TILED_CASE(2, 32, 4) // conv2 32 -> 64
    MZ_TILED_CASE(1, 64, 4) // residual blocks at C = 64 (24x24, 12x12)
    MZ_TILED_CASE(2, 32, 1) // C = 32 test nets: conv1 32 -> 16
    MZ_TILED_CASE(1, 16, 1)
    MZ_TILED_CASE(2, 16, 2) // conv2 16 -> 32
#undef MZ_TILED_CASE
    return launchConvAnyStrided(L, stride, params, in, skip, out, B, H, W, s); // any other width: the run-time-shaped kernel (net_wide.hip)
}

static DiscreteParams discreteParams(const float* p, const DiscreteHeadOffsets& o)
{
    return DiscreteParams{p + o.conv_w, p + o.conv_b, p + o.fc1_wT, p + o.fc1_b, p + o.fc2_wT, p + o.fc2_b, o.hc, o.hidden, o.size};
}

void Net::makeAtariHeadParams(AtariHeadParams* out) const
{
    AtariHeadParams& hp = *out;
    const float* params = params_.p;
    hp.reward = discreteParams(params, at_.reward);
    hp.value = discreteParams(params, at_.value);
    hp.pconv_w = params + heads_.pconv_w; hp.pconv_b = params + heads_.pconv_b; hp.pfc_wT = params + heads_.pfc_wT; hp.pfc_b = params + heads_.pfc_b;
    hp.C = desc_.num_hidden_channels; hp.P = P(); hp.A = desc_.action_size; hp.PC = heads_.pc;
}

static int launchAtariHeads(const Net& net, const float* params, const HeadOffsets& h, const AtariLayers& at, const float* x, int B, float* policy,
                            float* logit, float* value, float* reward, float* hidden_dst, const int* dst_idx, bool do_reward, hipStream_t s)
{
    AtariHeadParams hp;
    net.makeAtariHeadParams(&hp);
    const size_t lds = atariHeadsSmemFloats(hp) * sizeof(float);
    MZ_LDS_ATTR((heads_atari_kernel), lds);
    hipLaunchKernelGGL(heads_atari_kernel, dim3(B), dim3(1024), lds, s, x, hp, policy, logit, value, reward, hidden_dst, dst_idx, do_reward ? 1 : 0);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Net::initialAtari(const float* d_feat, int B, float* d_policy, float* d_logit, float* d_value, float* d_hidden, const int* d_dst_idx)
{
    MZ_HIP(hipSetDevice(device_));
    int rc = ensureBatch(B);
    if (rc) { return rc; }
    int H = desc_.input_channel_height, W = desc_.input_channel_width;
    const int C = desc_.num_hidden_channels;
    const size_t stage = size_t(C / 2) * (H / 2) * (W / 2); // floats per sample of the largest stage
    if (B > at_batch_) {
        MZ_HIP(hipStreamSynchronize(stream_));
        for (auto& b : at_buf_) { if (!b.alloc(size_t(B) * stage)) { setError("hipMalloc of the representation buffers failed"); return MZ_ERR_DEVICE; } }
        if (!at_out_.alloc(size_t(B) * hiddenSize())) { setError("hipMalloc of the representation buffers failed"); return MZ_ERR_DEVICE; }
        at_batch_ = B;
    }
    const float* P = params_.p;
    const int H0 = H, W0 = W;
    int Hout = 0, Wout = 0;
    // the layers of `n` samples from sample `first` on, on stream `s` (ref muzero_atari_network.py:21-39).  Every stage of these samples lives in
    // [first * stage, (first + n) * stage) of the three buffers, whatever the stage's size: two parts of the batch never touch each other's floats
    auto chain = [&](int first, int n, hipStream_t s) -> int {
        float *b0 = at_buf_[0].p + size_t(first) * stage, *b1 = at_buf_[1].p + size_t(first) * stage, *b2 = at_buf_[2].p + size_t(first) * stage;
        int h = H0, w = W0, rcc;
        auto pool = [&](const float* in, float* out) -> int {
            const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1, total = n * C * Ho * Wo;
            hipLaunchKernelGGL(avgpool3s2_kernel, dim3((total + 255) / 256), dim3(256), 0, s, in, C, h, w, out, total);
            MZ_HIP(hipGetLastError());
            h = Ho; w = Wo;
            return MZ_OK;
        };
        if ((rcc = launchTiled(at_.conv1, 2, P, d_feat + size_t(first) * featSize(), nullptr, b0, n, h, w, s, cu_count_))) { return rcc; }
        h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;
        if ((rcc = launchTiled(at_.rb1[0], 1, P, b0, nullptr, b1, n, h, w, s, cu_count_))) { return rcc; }
        if ((rcc = launchTiled(at_.rb1[1], 1, P, b1, b0, b2, n, h, w, s, cu_count_))) { return rcc; }
        if ((rcc = launchTiled(at_.conv2, 2, P, b2, nullptr, b0, n, h, w, s, cu_count_))) { return rcc; }
        h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;
        if ((rcc = launchTiled(at_.rb2[0], 1, P, b0, nullptr, b1, n, h, w, s, cu_count_))) { return rcc; }
        if ((rcc = launchTiled(at_.rb2[1], 1, P, b1, b0, b2, n, h, w, s, cu_count_))) { return rcc; }
        if ((rcc = pool(b2, b0))) { return rcc; }
        if ((rcc = launchTiled(at_.rb3[0], 1, P, b0, nullptr, b1, n, h, w, s, cu_count_))) { return rcc; }
        if ((rcc = launchTiled(at_.rb3[1], 1, P, b1, b0, b2, n, h, w, s, cu_count_))) { return rcc; }
        if ((rcc = pool(b2, at_out_.p + size_t(first) * hiddenSize()))) { return rcc; }
        Hout = h; Wout = w;
        return MZ_OK;
    };
    // A batch that fills the chip several times over runs as two halves on two streams.  The 4-5 workgroups a CU holds of one of these layers start together and
    // stay in step — all of them stage their patch, then all of them issue MFMAs, then all of them store (tools/ab_atari_root.sh: staging 10, MFMAs 28, epilogue
    // 11 of 48 us, nothing overlapping); the second half's launches arrive a phase later and fill what the first leaves idle.  Samples are independent: same outputs.
    static const int parts_env = getenv("MZ_REPR_PARTS") ? atoi(getenv("MZ_REPR_PARTS")) : 2; // (A/B switch; 1 = the whole batch on the network's stream)
    const int parts = std::max(1, std::min({parts_env, kReprParts, B / 16}));
    if (parts > 1) {
        if (!at_fork_) { // all or nothing: a failure half-way destroys what was made, so the next call starts from scratch instead of leaking the first set
            hipStream_t ns[kReprParts - 1] = {};
            hipEvent_t nj[kReprParts - 1] = {}, nf = nullptr;
            hipError_t e = hipSuccess;
            for (int k = 0; k < kReprParts - 1 && e == hipSuccess; ++k) { e = hipStreamCreateWithFlags(&ns[k], hipStreamNonBlocking); }
            for (int k = 0; k < kReprParts - 1 && e == hipSuccess; ++k) { e = hipEventCreateWithFlags(&nj[k], hipEventDisableTiming); }
            if (e == hipSuccess) { e = hipEventCreateWithFlags(&nf, hipEventDisableTiming); }
            if (e != hipSuccess) {
                for (hipStream_t q : ns) { if (q) { (void)hipStreamDestroy(q); } }
                for (hipEvent_t q : nj) { if (q) { (void)hipEventDestroy(q); } }
                setError("muzero_atari representation: side streams / events: %s", hipGetErrorString(e));
                return MZ_ERR_DEVICE;
            }
            for (int k = 0; k < kReprParts - 1; ++k) { at_streams_[k] = ns[k]; at_join_[k] = nj[k]; }
            at_fork_ = nf;
        }
        MZ_HIP(hipEventRecord(at_fork_, stream_)); // the features, and every earlier reader of the buffers
        // Whatever happens below, the side streams are joined before this function returns: a part that failed half-way must not leave another part's
        // kernels running on at_buf_ / at_out_ while the caller (or the next call's re-allocation, which only waits for stream_) goes on.
        int first_error = MZ_OK;
        bool forked[kReprParts] = {};
        for (int k = parts - 1; k >= 0 && first_error == MZ_OK; --k) {
            const int first = int(size_t(B) * k / parts), end = int(size_t(B) * (k + 1) / parts);
            hipStream_t s = k == 0 ? stream_ : at_streams_[k - 1];
            if (k > 0) {
                if (hipStreamWaitEvent(s, at_fork_, 0) != hipSuccess) { setError("muzero_atari representation: hipStreamWaitEvent failed"); first_error = MZ_ERR_DEVICE; break; }
                forked[k] = true;
            }
            if ((rc = chain(first, end - first, s))) { first_error = rc; }
        }
        for (int k = 1; k < parts; ++k) {
            if (!forked[k]) { continue; }
            if (first_error == MZ_OK) {
                if (hipEventRecord(at_join_[k - 1], at_streams_[k - 1]) != hipSuccess || hipStreamWaitEvent(stream_, at_join_[k - 1], 0) != hipSuccess) {
                    setError("muzero_atari representation: joining a side stream failed");
                    first_error = MZ_ERR_DEVICE;
                }
            }
            if (first_error != MZ_OK) { (void)hipStreamSynchronize(at_streams_[k - 1]); } // error path: a plain wait (the error message of the first failure stays)
        }
        if (first_error != MZ_OK) { return first_error; }
    } else if ((rc = chain(0, B, stream_))) {
        return rc;
    }
    H = Hout; W = Wout;
    if (H != desc_.hidden_channel_height || W != desc_.hidden_channel_width) { setError("internal: representation output %dx%d", H, W); return MZ_ERR_ARG; }
    float* b0 = at_out_.p;
    const float* x = b0;
    if (!at_.tail.empty()) {
        bool launched = false;
        if ((rc = launchTower(at_.tail, b0, act_[0].p, B, &launched, false, false))) { return rc; }
        if (launched) { x = act_[0].p; }
        else { // any other width: residual blocks layer by layer (launchConv ends in the run-time-shaped kernel)
            const float* xin = b0;
            float *tmp = act_[1].p, *y = act_[0].p, *y2 = act_[2].p;
            for (size_t i = 0; i + 1 < at_.tail.size(); i += 2) { // ref network_unit.py:14-23
                if ((rc = launchConv(at_.tail[i], xin, nullptr, tmp, B))) { return rc; }
                if ((rc = launchConv(at_.tail[i + 1], tmp, xin, y, B))) { return rc; }
                xin = y;
                float* s = y; y = y2; y2 = s;
            }
            x = xin;
        }
    }
    return launchAtariHeads(*this, P, heads_, at_, x, B, d_policy, d_logit, d_value, nullptr, d_hidden, d_dst_idx, false, stream_);
}

// dynamics input for Atari: cat(hidden[src], 18 action planes with plane `action` all ones) (ref atari.cpp:124-130)
__global__ void build_recurrent_input_atari(const float* __restrict__ hidden, const int* __restrict__ src_idx, const float* __restrict__ planes,
                                            const int* __restrict__ action_ids, int C, int AC, int P, float* __restrict__ out)
{
    const int b = blockIdx.x;
    const int s = src_idx ? src_idx[b] : b;
    const float* h = hidden + size_t(s) * C * P;
    float* o = out + size_t(b) * (C + AC) * P;
    for (int i = threadIdx.x; i < C * P; i += blockDim.x) { o[i] = h[i]; }
    for (int i = threadIdx.x; i < AC * P; i += blockDim.x) {
        o[C * P + i] = planes ? planes[size_t(b) * AC * P + i] : ((i / P) == action_ids[b] ? 1.0f : 0.0f);
    }
}

int Net::recurrentAtari(const float* d_hidden_src, const int* d_src_idx, const float* d_action_planes, const int* d_action_ids, int B, float* d_policy,
                        float* d_logit, float* d_value, float* d_reward, float* d_hidden_dst, const int* d_dst_idx)
{
    MZ_HIP(hipSetDevice(device_));
    int rc = ensureBatch(B);
    if (rc) { return rc; }
    hipLaunchKernelGGL(build_recurrent_input_atari, dim3(B), dim3(256), 0, stream_, d_hidden_src, d_src_idx, d_action_planes, d_action_ids,
                       desc_.num_hidden_channels, desc_.num_action_feature_channels, P(), rec_in_.p);
    MZ_HIP(hipGetLastError());
    bool launched = false;
    if ((rc = launchTower(dyn_, rec_in_.p, act_[0].p, B, &launched, false, true))) { return rc; }
    float* xo = act_[0].p;
    if (!launched && (rc = runTrunk(dyn_, rec_in_.p, B, &xo))) { return rc; } // any other width: layer by layer
    if (conv_only_) { return MZ_OK; }
    return launchAtariHeads(*this, params_.p, heads_, at_, xo, B, d_policy, d_logit, d_value, d_reward, d_hidden_dst, d_dst_idx, true, stream_);
}

__global__ void invert_value_kernel(const float* __restrict__ in, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[i] = invertValueDev(in[i]); }
}

int invertValuesOnDevice(int device, const float* values, int n, float* out)
{
    if (n <= 0) { return MZ_OK; }
    MZ_HIP(hipSetDevice(device));
    DevBuf<float> d_in, d_out;
    if (!d_in.alloc(n) || !d_out.alloc(n)) { setError("invertValuesOnDevice: allocation failed"); return MZ_ERR_DEVICE; }
    MZ_HIP(hipMemcpy(d_in.p, values, size_t(n) * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(invert_value_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, d_in.p, n, d_out.p);
    MZ_HIP(hipGetLastError());
    MZ_HIP(hipMemcpy(out, d_out.p, size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
    return MZ_OK;
}

// raw observations -> the network's float planes (ref atari.cpp:199-216 getFeatures: per history step one action plane, then R, G, B / 255):
// `raw` per sample = hist screens oldest first (3 * res * res bytes each), hist f32 action-plane values, hist valid bytes (env.cpp rawFeatures).
// Same values as the host's features() (one IEEE division per element); the point is the 4x smaller host-to-device copy per move.
__global__ __launch_bounds__(256) void atari_expand_features(const uint8_t* __restrict__ raw, int raw_bytes, int hist, int res, float* __restrict__ out)
{
    const int b = blockIdx.y, i = blockIdx.x; // sample, history step
    const int pix = res * res, frame = 3 * pix;
    const uint8_t* r = raw + size_t(b) * raw_bytes;
    float av;
    memcpy(&av, r + size_t(hist) * frame + size_t(i) * 4, 4);
    const bool valid = r[size_t(hist) * frame + size_t(hist) * 4 + i] != 0;
    const uint8_t* f = r + size_t(i) * frame;
    float* dst = out + (size_t(b) * hist + i) * 4 * pix;
    for (int p = threadIdx.x; p < pix; p += 256) { dst[p] = av; }
    for (int p = threadIdx.x; p < frame; p += 256) { dst[pix + p] = valid ? static_cast<float>(f[p]) / 255.0f : 0.0f; }
}

// the same from the PREVIOUS move's raw block shifted by one screen + the newest screen (uploaded alone) + the new (action values, valid flags):
// writes the new raw block (`cur`) and the float planes; one block per (history step, sample)
__global__ __launch_bounds__(256) void atari_shift_expand_features(const uint8_t* __restrict__ prev, const uint8_t* __restrict__ newest, const uint8_t* __restrict__ meta,
                                                                   uint8_t* __restrict__ cur, int raw_bytes, int hist, int res, float* __restrict__ out)
{
    const int b = blockIdx.y, i = blockIdx.x;
    const int pix = res * res, frame = 3 * pix;
    const uint8_t* m = meta + size_t(b) * hist * 5;
    float av;
    memcpy(&av, m + size_t(i) * 4, 4);
    const bool valid = m[size_t(hist) * 4 + i] != 0;
    const uint8_t* f = i + 1 < hist ? prev + size_t(b) * raw_bytes + size_t(i + 1) * frame : newest + size_t(b) * frame;
    uint8_t* c = cur + size_t(b) * raw_bytes + size_t(i) * frame;
    float* dst = out + (size_t(b) * hist + i) * 4 * pix;
    // byte / 255.0f for the 256 byte values once per block (the same IEEE division, ref atari.cpp:112-122), then four bytes per load and one 16-byte store per
    // thread and step (a byte per load and a division per element: 48 us per 64 roots, 75 MB of planes at 1.5 TB/s)
    __shared__ float tab[256];
    tab[threadIdx.x] = valid ? static_cast<float>(threadIdx.x) / 255.0f : 0.0f;
    __syncthreads();
    if ((pix & 3) == 0 && (reinterpret_cast<uintptr_t>(f) & 3) == 0 && (reinterpret_cast<uintptr_t>(c) & 3) == 0) {
        const float4 a4 = make_float4(av, av, av, av);
        for (int p = threadIdx.x; p < pix / 4; p += 256) { reinterpret_cast<float4*>(dst)[p] = a4; }
        const uchar4* f4 = reinterpret_cast<const uchar4*>(f);
        uchar4* c4 = reinterpret_cast<uchar4*>(c);
        float4* d4 = reinterpret_cast<float4*>(dst + pix);
        for (int p = threadIdx.x; p < frame / 4; p += 256) {
            const uchar4 v = f4[p];
            c4[p] = v;
            d4[p] = make_float4(tab[v.x], tab[v.y], tab[v.z], tab[v.w]);
        }
    } else {
        for (int p = threadIdx.x; p < pix; p += 256) { dst[p] = av; }
        for (int p = threadIdx.x; p < frame; p += 256) {
            const uint8_t v = f[p];
            c[p] = v;
            dst[pix + p] = tab[v];
        }
    }
    if (i == 0) { for (int p = threadIdx.x; p < hist * 5; p += 256) { cur[size_t(b) * raw_bytes + size_t(hist) * frame + p] = m[p]; } }
}

int Net::shiftExpandAtariFeatures(const uint8_t* d_prev, const uint8_t* d_newest, const uint8_t* d_meta, uint8_t* d_cur, int raw_bytes, int B, float* d_feat)
{
    const int res = desc_.input_channel_height, hist = desc_.num_input_channels / 4;
    if (desc_.input_channel_width != res || raw_bytes != hist * 3 * res * res + hist * 5) { setError("shiftExpandAtariFeatures: unexpected observation layout"); return MZ_ERR_ARG; }
    hipLaunchKernelGGL(atari_shift_expand_features, dim3(hist, B), dim3(256), 0, stream_, d_prev, d_newest, d_meta, d_cur, raw_bytes, hist, res, d_feat);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Net::expandAtariFeatures(const uint8_t* d_raw, int raw_bytes, int B, float* d_feat)
{
    const int res = desc_.input_channel_height, hist = desc_.num_input_channels / 4;
    if (desc_.input_channel_width != res || raw_bytes != hist * 3 * res * res + hist * 5) { setError("expandAtariFeatures: unexpected observation layout"); return MZ_ERR_ARG; }
    hipLaunchKernelGGL(atari_expand_features, dim3(hist, B), dim3(256), 0, stream_, d_raw, raw_bytes, hist, res, d_feat);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

} // namespace mz
