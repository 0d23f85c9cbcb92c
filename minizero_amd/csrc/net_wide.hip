// Network kernels for the shapes beyond BASELINE.json (gfx950, -ffp-contract=off; arithmetic order: DESIGN.md §4):
//  tower_wide   — the whole trunk of one sample per workgroup with ONE activation tile in LDS (net_wide_body.h): 128 / 256 hidden channels (the
//                 reference's default network is 1 block x 256 channels, config/configuration.cpp:70-72), 7x7 / 13x13 / 19x19 Go (go_unit.h:11)
//  conv3x3_any  — a 3x3 convolution of ANY shape (run-time H, W, channels): the instance behind everything else, so that no network the reference's
//                 create_network.py can build (network/py/create_network.py, network_unit.py:6-87) ends in "no kernel instance".  The B operand is read
//                 straight from global memory (L1 / L2) with the window's border test per lane; the chain per output is the contract's (tap-major,
//                 channel ascending), an MFMA step with a zero operand leaving the chain's value as it is.
#include "net.h"
#include "net_wide_body.h"
#include <cstring>

namespace mz {

template <int H, int W, int CIN0Q, int C>
__global__ __launch_bounds__(512) void tower_wide(const float* __restrict__ in, const float* __restrict__ params, TowerArgs ta, float* __restrict__ out,
                                                  float* __restrict__ tmp)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr size_t CP = size_t(C) * H * W;
    wideTowerBody<H, W, CIN0Q, C>(in, params, ta, out + blockIdx.x * CP, tmp + blockIdx.x * CP, blockIdx.x, threadIdx.x, tile, false);
}

// grid (B, pixel chunks of 64): wave w of the 4 loops over the oc-tiles w, w + 4, ...; 4 pixel tiles per block
__global__ __launch_bounds__(256) void conv3x3_any(const float* __restrict__ in, int cin, int CG, const float* __restrict__ wp, const float* __restrict__ bias,
                                                   const float* __restrict__ skip, float* __restrict__ out, int cout, int OT, int H, int W)
{
    const int P = H * W, b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kc = lane >> 4;
    const float* src = in + size_t(b) * cin * P;
    int q[4], pos[4];
    unsigned mask[4]; // bit t: tap t of pixel j is inside the board
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q[j] = 64 * blockIdx.y + 16 * j + (lane & 15);
        const bool ok = q[j] < P;
        const int y = ok ? q[j] / W : 0, x = ok ? q[j] % W : 0;
        pos[j] = y * W + x;
        unsigned m = 0;
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (ok && yy >= 0 && yy < H && xx >= 0 && xx < W) { m |= 1u << t; }
        }
        mask[j] = m;
    }
    for (int ot = wave; ot < OT; ot += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
        for (int t = 0; t < 9; ++t) {
            const int d = (t / 3 - 1) * W + (t % 3 - 1);
            for (int cg = 0; cg < CG; ++cg) {
                const float a = wp[((size_t(t) * CG + cg) * OT + ot) * 64 + lane];
                const int c = 4 * cg + kc;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float bv = 0.0f;
                    if (c < cin && ((mask[j] >> t) & 1u)) { bv = src[size_t(c) * P + pos[j] + d]; }
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[j], 0, 0, 0);
                }
            }
        }
        float* dst = out + size_t(b) * cout * P;
        const float* sk = skip ? skip + size_t(b) * cout * P : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oc = 16 * ot + 4 * kc + r;
                if (q[j] < P && oc < cout) {
                    float v = acc[j][r] + bias[oc];
                    if (sk) { v = v + sk[size_t(oc) * P + q[j]]; }
                    dst[size_t(oc) * P + q[j]] = v > 0.0f ? v : 0.0f;
                }
            }
        }
    }
}

// ... and with a stride and rectangular planes (the stride-2 stem convs and the 48x48 .. 12x12 residual blocks of a muzero_atari representation whose width has
// no conv3x3_tiled instance, ref muzero_atari_network.py:21-70): output (oy, ox) reads input (oy * stride + dy, ox * stride + dx), pad 1
__global__ __launch_bounds__(256) void conv3x3_any_strided(const float* __restrict__ in, int cin, int CG, const float* __restrict__ wp, const float* __restrict__ bias,
                                                           const float* __restrict__ skip, float* __restrict__ out, int cout, int OT, int H, int W, int stride)
{
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1, Po = Ho * Wo, Pi = H * W;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kc = lane >> 4;
    const float* src = in + size_t(b) * cin * Pi;
    int q[4], pos[4];
    unsigned mask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q[j] = 64 * blockIdx.y + 16 * j + (lane & 15);
        const bool ok = q[j] < Po;
        const int y = ok ? (q[j] / Wo) * stride : 0, x = ok ? (q[j] % Wo) * stride : 0;
        pos[j] = y * W + x;
        unsigned m = 0;
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (ok && yy >= 0 && yy < H && xx >= 0 && xx < W) { m |= 1u << t; }
        }
        mask[j] = m;
    }
    for (int ot = wave; ot < OT; ot += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
        for (int t = 0; t < 9; ++t) {
            const int d = (t / 3 - 1) * W + (t % 3 - 1);
            for (int cg = 0; cg < CG; ++cg) {
                const float a = wp[((size_t(t) * CG + cg) * OT + ot) * 64 + lane];
                const int c = 4 * cg + kc;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float bv = 0.0f;
                    if (c < cin && ((mask[j] >> t) & 1u)) { bv = src[size_t(c) * Pi + pos[j] + d]; }
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[j], 0, 0, 0);
                }
            }
        }
        float* dst = out + size_t(b) * cout * Po;
        const float* sk = skip ? skip + size_t(b) * cout * Po : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oc = 16 * ot + 4 * kc + r;
                if (q[j] < Po && oc < cout) {
                    float v = acc[j][r] + bias[oc];
                    if (sk) { v = v + sk[size_t(oc) * Po + q[j]]; }
                    dst[size_t(oc) * Po + q[j]] = v > 0.0f ? v : 0.0f;
                }
            }
        }
    }
}

// one bit per point -> f32 0 / 1 planes (the lock-step worker stages bit-packed planes; the run-time-shaped kernels read floats)
__global__ __launch_bounds__(256) void unpack_bits_kernel(const unsigned* __restrict__ bits, int C, int P, float* __restrict__ feat)
{
    const int b = blockIdx.x, W32 = (P + 31) / 32;
    for (int i = threadIdx.x; i < C * P; i += 256) {
        const int c = i / P, p = i - c * P;
        feat[size_t(b) * C * P + i] = ((bits[(size_t(b) * C + c) * W32 + (p >> 5)] >> (p & 31)) & 1u) ? 1.0f : 0.0f;
    }
}

// ---- host side ----
#define MZ_WIDE_CASES(X) /* (H, W, input channels of layer 0 padded to 16, hidden channels) */ \
    X(9, 9, 32, 128)   /* 9x9 Go, 128 channels */ \
    X(9, 9, 32, 256)   /* 9x9 Go, 256 channels: the reference's default width (configuration.cpp:71) */ \
    X(9, 9, 32, 32)    \
    X(7, 7, 32, 32)    /* 7x7 Go (docs/Training.md trains it) */ \
    X(7, 7, 32, 64)    \
    X(7, 7, 32, 128)   \
    X(7, 7, 32, 256)   \
    X(13, 13, 32, 64)  /* 13x13 Go */ \
    X(13, 13, 32, 128) \
    X(19, 19, 32, 32)  /* 19x19 Go */ \
    X(19, 19, 32, 64)  \
    X(9, 9, 144, 128)  /* MuZero dynamics: hidden + one action plane */ \
    X(9, 9, 272, 256)  \
    X(7, 7, 80, 64)    \
    X(13, 13, 80, 64)  \
    X(19, 19, 80, 64)

template <int H, int W, int CIN0Q, int C>
static int launchTowerWideT(const TowerArgs& ta, const float* params, const float* in, float* out, float* tmp, int B, hipStream_t s)
{
    constexpr size_t lds = wideTileFloats<H, W, C>(CIN0Q) * sizeof(float);
    static_assert(lds <= 160 * 1024, "the wide tower's tile must fit the LDS");
    MZ_LDS_ATTR((tower_wide<H, W, CIN0Q, C>), lds);
    hipLaunchKernelGGL((tower_wide<H, W, CIN0Q, C>), dim3(B), dim3(512), lds, s, in, params, ta, out, tmp);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

// TowerArgs of a trunk for the wide tower (wq offsets); false: not the shape it handles.  *c0q = input channels of layer 0 padded to 16
bool Net::makeWideArgs(const std::vector<ConvLayer>& t, bool in_bits, TowerArgs* out, int* c0q) const
{
    if (!use_fused_ || t.size() > 48 || t.empty() || (t.size() % 2) == 0) { return false; }
    const int C = desc_.num_hidden_channels;
    if (C % 16 != 0) { return false; }
    for (size_t i = 0; i < t.size(); ++i) { if ((i > 0 && t[i].cin != C) || t[i].cout != C) { return false; } }
    TowerArgs& ta = *out;
    memset(&ta, 0, sizeof(ta));
    ta.nlayers = static_cast<int>(t.size());
    ta.cin0 = t[0].cin;
    ta.C = C;
    ta.OT = C / 16;
    ta.in_bits = in_bits ? 1 : 0;
    ta.has_stem = 1;
    for (size_t i = 0; i < t.size(); ++i) { ta.w_off[i] = static_cast<unsigned>(t[i].wq_off); ta.b_off[i] = static_cast<unsigned>(t[i].b_off); }
    *c0q = 16 * t[0].cq;
    return true;
}

bool Net::hasWideTower(const std::vector<ConvLayer>& t) const
{
    TowerArgs ta;
    int c0q = 0;
    if (!makeWideArgs(t, false, &ta, &c0q)) { return false; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
#define MZ_WIDE_HAS(h, w, cin0q, c) \
    if (H == h && W == w && c0q == cin0q && C == c) { return true; }
    MZ_WIDE_CASES(MZ_WIDE_HAS)
#undef MZ_WIDE_HAS
    return false;
}

int Net::launchTowerWide(const std::vector<ConvLayer>& t, const float* in, float* out, float* tmp, int B, bool* launched, bool in_bits)
{
    *launched = false;
    TowerArgs ta;
    int c0q = 0;
    if (!makeWideArgs(t, in_bits, &ta, &c0q)) { return MZ_OK; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
#define MZ_WIDE_LAUNCH(h, w, cin0q, c) \
    if (H == h && W == w && c0q == cin0q && C == c) { *launched = true; return launchTowerWideT<h, w, cin0q, c>(ta, params_.p, in, out, tmp, B, stream_); }
    MZ_WIDE_CASES(MZ_WIDE_LAUNCH)
#undef MZ_WIDE_LAUNCH
    return MZ_OK;
}

int Net::launchConvAny(const ConvLayer& L, const float* in, const float* skip, float* out, int B)
{
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width;
    hipLaunchKernelGGL(conv3x3_any, dim3(B, (H * W + 63) / 64), dim3(256), 0, stream_, in, L.cin, L.cin_pad / 4, params_.p + L.w_off, params_.p + L.b_off, skip, out, L.cout,
                       L.cout_pad / 16, H, W);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int launchConvAnyStrided(const ConvLayer& L, int stride, const float* params, const float* in, const float* skip, float* out, int B, int H, int W, hipStream_t s)
{
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    hipLaunchKernelGGL(conv3x3_any_strided, dim3(B, (Ho * Wo + 63) / 64), dim3(256), 0, s, in, L.cin, L.cin_pad / 4, params + L.w_off, params + L.b_off, skip, out, L.cout,
                       L.cout_pad / 16, H, W, stride);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int Net::unpackBits(const float* d_bits, int C, int B, float* d_feat)
{
    hipLaunchKernelGGL(unpack_bits_kernel, dim3(B), dim3(256), 0, stream_, reinterpret_cast<const unsigned*>(d_bits), C, P(), d_feat);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

} // namespace mz
