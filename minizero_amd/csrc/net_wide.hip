// Network kernels for the shapes beyond BASELINE.json (gfx950, -ffp-contract=off; arithmetic order: DESIGN.md §4):
//  tower_wide   — the whole trunk of one sample per workgroup with ONE activation tile in LDS (net_wide_body.h): 128 / 256 hidden channels (the
//                 reference's default network is 1 block x 256 channels, config/configuration.cpp:70-72), 7x7 / 13x13 / 19x19 Go (go_unit.h:11)
//  conv3x3_band — a 3x3 convolution of ANY shape (run-time H, W, channels) with its B operand in LDS: a workgroup stages a band of output rows of one sample and runs
//                 the layer out of it like the towers do.  The instance behind everything else, so that no network the reference's create_network.py can build
//                 (network/py/create_network.py, network_unit.py:6-87) ends in "no kernel instance" — and does not crawl either (0.5-0.6 of the f32-MFMA peak).
//  conv3x3_any  — its predecessor and fallback (layers of more than ~2000 input channels; MZ_NO_CONV_BAND=1): the B operand read straight from global memory
//                 (L1 / L2) with the window's border test per lane.  Both: the chain per output is the contract's (tap-major, channel ascending), an MFMA step
//                 with a zero operand leaving the chain's value as it is.
#include "net.h"
#include "net_wide_body.h"
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <cmath>
#include <cstdlib>
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

// conv3x3_band — the run-time-shaped convolution with its B operand in LDS (round 5, behind every shape without a fused instance: 19x19 x 128 / 256, 13x13 x 96,
// 11x11 x 48, ...).  conv3x3_any reads every B value from global memory, five loads per MFMA: 17-35 TFLOP/s.  Here a workgroup takes a BAND of TH output rows of one sample
// (grid = samples x bands; TH is chosen on the host so that the band's padded patch of ALL input channels fits the LDS — the contract's chain is tap-major, so a
// channel-chunked patch would have to be staged nine times), stages the (TH + 2) x (W + 2) patch once and runs the layer out of it like the towers do: pixel tiles are 16
// consecutive pixels of the band, a JOB is one oc-tile x up to 6 pixel tiles (accumulators that share the A fragment), the 8 waves take the jobs round-robin.  A fragments
// (weights.cpp `wp` layout: [tap][channel group][oc-tile][64], i.e. linear in the flattened step) travel in chunks of 8 k-steps, double-buffered; the B values of step
// s + 1 are read before the MFMAs of step s are issued.  Same chain per output as everywhere: tap-major, channels ascending, zero operands for the border and the padding.
template <int NT>
__device__ __forceinline__ void bandJob(const float* __restrict__ xs, const float* __restrict__ wl, int CG, int OT, int CS, int PW, const int (&lb)[6], f32x4 (&acc)[6])
{
    // The layer's k-steps flattened: s = tap * CG + channel group, S = 9 * CG of them; the A fragment of step s lies at wl[s * OT * 64] (weights.cpp `wp`), so chunks of
    // 8 steps need no tap boundaries.  The main loop's body is straight-line code — 8 steps, no branch (the first version guarded every step with `cg < CG` and the
    // compiler's conservative waits at the joins made every MFMA wait for the LDS read issued just before it: 0.43 of peak at 19x19 x 256) —, the S % 8 last steps run one by one.
    const int S = 9 * CG, nfull = S >> 3;
    const size_t astep = size_t(OT) * 64;
    float a_cur[8], a_nxt[8], bc[NT];
    // (t, cg) of the step whose B values are being FETCHED, as the LDS offset o = tap offset + 4 * cg * CS, advanced with scalar selects
    int cg = 0, tcol = 0, rowoff = 0, o = 0;
    auto advance = [&]() {
        ++cg;
        o += 4 * CS;
        if (cg == CG) {
            cg = 0;
            ++tcol;
            if (tcol == 3) { tcol = 0; rowoff += PW; }
            o = rowoff + tcol;
        }
    };
    auto bload = [&](float (&b)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) { b[j] = xs[lb[j] + o]; }
    };
    {
        float b0[NT];
        bload(b0);
        // (through a VALU move: a bc that is the direct target of an LDS read at the loop's entry keeps the compiler's wait-count analysis pessimistic inside the loop)
#pragma unroll
        for (int j = 0; j < NT; ++j) { asm volatile("v_mov_b32 %0, %1" : "=v"(bc[j]) : "v"(b0[j])); }
    }
    // A fragments travel TWO chunks ahead of their MFMAs (one chunk = 8 x NT x 32 cycles of the wave's own issue: not always an L2 round trip under load) in three
    // register sets whose roles rotate — the loop body is three chunks, no set is ever copied (a copy waits for the loads it copies)
    float a_nn[8];
    auto loadA = [&](float (&a)[8], int s0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = wl[size_t(s0 + u < S ? s0 + u : S - 1) * astep]; } // (no branch around a load: beyond the layer, its last fragment again)
    };
    auto chunk = [&](const float (&a_use)[8], float (&a_load)[8], int i) {
        loadA(a_load, 8 * (i + 2));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float bn[NT];
            if (8 * i + u + 1 < S) { advance(); } // (the layer's very last step fetches its own operands again)
            bload(bn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_use[u], bc[j], acc[j], 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) { bc[j] = bn[j]; }
        }
    };
    loadA(a_cur, 0);
    loadA(a_nxt, 8);
    int i = 0;
#pragma unroll 1
    for (; i + 3 <= nfull; i += 3) {
        chunk(a_cur, a_nn, i);
        chunk(a_nxt, a_cur, i + 1);
        chunk(a_nn, a_nxt, i + 2);
    }
    if (i < nfull) { // one or two chunks are left (roles as at the top of the loop body)
        chunk(a_cur, a_nn, i);
        if (i + 1 < nfull) { chunk(a_nxt, a_cur, i + 1); }
    }
#pragma unroll 1
    for (int s = 8 * nfull; s < S; ++s) { // bc = the operands of step s (fetched by the step before)
        const float a = wl[size_t(s) * astep];
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[j], acc[j], 0, 0, 0); }
        if (s + 1 < S) { advance(); bload(bc); }
    }
}

__global__ __launch_bounds__(1024) void conv3x3_band(const float* __restrict__ in, int cin, int CG, const float* __restrict__ wp, const float* __restrict__ bias,
                                                    const float* __restrict__ skip, float* __restrict__ out, int cout, int OT, int H, int W, int TH, int CS, int GM)
{
    extern __shared__ __attribute__((aligned(16))) float xs[]; // [4 * CG][CS]: the band's padded patch, channel-major
    const int b = blockIdx.x, r0 = blockIdx.y * TH, th = H - r0 < TH ? H - r0 : TH;
    const int tid = threadIdx.x, lane = tid & 63, kc = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave-uniform for the compiler too: the job, its oc-tile and its tile count are scalars)
    const int P = H * W, PW = W + 2, PP = (th + 2) * PW, cin_pad = 4 * CG;
    const float* src = in + size_t(b) * cin * P;
    // the patch: a thread keeps ONE position of the padded plane (bounds test and source offset once) and walks a slice of the channels, eight loads in flight; a band has
    // few positions (19x19, TH = 4: 126) and many channels, so the workgroup's threads are G = threads / PP groups of PP threads and group k takes the channel batches k, k + G, ...
    // (one group walking all 128 channels was 16 dependent trips to the L2 per layer: 16 of a layer's 125 us)
    {
        const int NTH = blockDim.x, G = PP < NTH ? NTH / PP : 1, grp = tid / PP;
        for (int pos = G > 1 ? tid - grp * PP : tid; pos < PP && grp < G; pos += NTH) {
            const int r = pos / PW, q = pos - r * PW;
            const int iy = r0 + r - 1, ix = q - 1;
            const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float* p = src + (inside ? iy * W + ix : 0);
            for (int c0 = 8 * grp; c0 < cin_pad; c0 += 8 * G) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = inside && c0 + u < cin;
                    const float x = p[ok ? size_t(c0 + u) * P : 0];
                    v[u] = ok ? x : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { if (c0 + u < cin_pad) { xs[(c0 + u) * CS + pos] = v[u]; } }
            }
        }
    }
    __syncthreads();
    const int npix = th * W, ntiles = (npix + 15) >> 4, npg = (ntiles + GM - 1) / GM, njobs = OT * npg; // GM <= 6: pixel tiles per job (host: planBand)
    float* dst = out + size_t(b) * cout * P + r0 * W;
    const float* sk = skip ? skip + size_t(b) * cout * P + r0 * W : nullptr;
    const int nwaves = blockDim.x >> 6; // 8, or 16 where the patch leaves room for one workgroup per CU only (host: planBand)
    for (int job = wave; job < njobs; job += nwaves) {
        const int ot = job / npg, pg = job - ot * npg;
        const int nt = ntiles - GM * pg < GM ? ntiles - GM * pg : GM; // pixel tiles of this job (wave-uniform)
        int lb[6], pq[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int q = (GM * pg + j) * 16 + (lane & 15);
            const bool ok = j < nt && q < npix;
            const int y = ok ? q / W : 0, x = ok ? q - y * W : 0;
            lb[j] = kc * CS + y * PW + x; // top-left tap of the pixel's window in the padded patch, channel kc of a group
            pq[j] = ok ? q : -1;
        }
        f32x4 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
        const float* wl = wp + size_t(ot) * 64 + lane;
        switch (nt) {
            case 6: bandJob<6>(xs, wl, CG, OT, CS, PW, lb, acc); break;
            case 5: bandJob<5>(xs, wl, CG, OT, CS, PW, lb, acc); break;
            case 4: bandJob<4>(xs, wl, CG, OT, CS, PW, lb, acc); break;
            case 3: bandJob<3>(xs, wl, CG, OT, CS, PW, lb, acc); break;
            case 2: bandJob<2>(xs, wl, CG, OT, CS, PW, lb, acc); break;
            default: bandJob<1>(xs, wl, CG, OT, CS, PW, lb, acc); break;
        }
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int oc = 16 * ot + 4 * kc + r; bv[r] = bias[oc < cout ? oc : 0]; }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (pq[j] < 0) { continue; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oc = 16 * ot + 4 * kc + r;
                if (oc < cout) {
                    float v = acc[j][r] + bv[r];
                    if (sk) { v = v + sk[size_t(oc) * P + pq[j]]; }
                    dst[size_t(oc) * P + pq[j]] = v > 0.0f ? v : 0.0f;
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
    X(8, 8, 16, 128)   /* Othello (4 input planes) at the reference's default width and half of it */ \
    X(8, 8, 16, 256)   \
    X(3, 3, 16, 128)   /* TicTacToe: docs/Training.md's first example trains it with the default 1 block x 256 channels */ \
    X(3, 3, 16, 256)   \
    X(9, 9, 144, 128)  /* MuZero dynamics: hidden + one action plane */ \
    X(9, 9, 272, 256)  \
    X(7, 7, 80, 64)    \
    X(13, 13, 80, 64)  \
    X(19, 19, 80, 64)  \
    X(8, 8, 272, 256)  /* MuZero dynamics on the Othello / TicTacToe boards */ \
    X(8, 8, 144, 128)  \
    X(3, 3, 272, 256)

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

// The band height, the pixel tiles per job and the waves per workgroup of conv3x3_band for a layer shape and a batch: the (TH, GM) with the shortest estimated launch —
// rounds of workgroups over the CUs x (MFMA issue of a band's slowest SIMD, in pixel tiles x the layer's steps, + staging + fixed costs) — among the bands whose patch
// fits the LDS.  TH = 0: no band fits (more than ~2000 input channels).
struct BandPlan { int TH = 0, CS = 0, GM = 6, NW = 8; size_t lds = 0; };
static BandPlan planBand(int H, int W, int cin_pad, int OT, int B, int cus)
{
    BandPlan best;
    double best_cost = 0.0;
    for (int TH = 1; TH <= H; ++TH) {
        const int PP = (TH + 2) * (W + 2);
        const int CS = PP + ((16 - PP % 32) + 32) % 32; // >= PP, % 32 == 16 (net_dev.h planeStride: the four channels of a k-group on different banks)
        const size_t lds = size_t(cin_pad) * CS * sizeof(float);
        if (lds > size_t(156) * 1024) { break; }
        const int nb = (H + TH - 1) / TH;
        // 16 waves per CU hide what 8 do not (19x19 x 128: two workgroups of 8 waves with bands of 4 rows 0.59 of peak, one with bands of 10 rows 0.50): a patch that
        // leaves room for one workgroup per CU gets 16 waves in it
        const int NW = lds > size_t(78) * 1024 ? 16 : 8;
        const int slots = NW == 16 ? 1 : 2; // workgroups per CU (16 waves)
        const double rounds = std::ceil(double(B) * nb / (double(std::max(1, cus)) * slots));
        for (int GM = 6; GM >= 2; --GM) { // pixel tiles per job: fewer = more jobs for the waves (narrow layers), more = fewer A fragments fetched
            double mfma = 0.0, fixed = 0.0;
            for (int k = 0; k < nb; ++k) {
                const int th = std::min(TH, H - k * TH), ntiles = (th * W + 15) / 16, npg = (ntiles + GM - 1) / GM;
                int load[4] = {0, 0, 0, 0}; // pixel tiles per SIMD (waves w, w + 4, ... share one)
                for (int job = 0; job < OT * npg; ++job) { load[(job % NW) % 4] += std::min(GM, ntiles - GM * (job % npg)); }
                const int mx = std::max(std::max(load[0], load[1]), std::max(load[2], load[3]));
                mfma += double(mx) * 9.0 * (cin_pad / 4) * 32.0;                                                              // MFMA issue of the slowest SIMD
                fixed += double(cin_pad) * (th + 2) * (W + 2) / 8.0 + 12000.0 + 5000.0 * ((OT * npg + NW - 1) / NW);           // staging (~32 B per cycle and CU), barrier, the jobs' prologues and epilogues
            }
            // workgroups that share a CU share its MFMA pipes; their fixed parts overlap
            const double cost = rounds * (slots * mfma + fixed) / nb;
            if (best.TH == 0 || cost < best_cost) { best.TH = TH; best.CS = CS; best.GM = GM; best.NW = NW; best.lds = lds; best_cost = cost; }
        }
    }
    return best;
}

// conv3x3_band for a stride-1 layer (weights.cpp `wp` layout at L.w_off); *launched = false: no band fits, nothing was launched
int launchConvBand(const ConvLayer& L, const float* params, const float* in, const float* skip, float* out, int B, int H, int W, hipStream_t s, int cus, bool* launched)
{
    *launched = false;
    if (B <= 0) { *launched = true; return MZ_OK; } // (an empty batch: nothing to launch, and nothing for the caller to fall back to)
    // (the plan of a layer shape and batch is remembered: the search over band heights is tens of microseconds of host time, a lock-step cycle asks for 13 of them)
    BandPlan bp;
    {
        // a few remembered plans per host thread in front of the shared table (no lock on the per-cycle path: a trunk asks for two keys, stem and body, thirteen
        // times a cycle); the shared table is bounded — a caller that varies B without end starts over instead of growing it
        const std::array<int, 6> key{H, W, L.cin_pad, L.cout_pad / 16, B, cus};
        struct Recent { std::array<int, 6> key; BandPlan plan; bool valid = false; };
        thread_local Recent recent[4];
        thread_local int recent_next = 0;
        bool hit = false;
        for (const Recent& r : recent) { if (r.valid && r.key == key) { bp = r.plan; hit = true; break; } }
        if (!hit) {
            static std::mutex mu;
            static std::map<std::array<int, 6>, BandPlan> plans;
            {
                std::lock_guard<std::mutex> lock(mu);
                auto it = plans.find(key);
                if (it == plans.end()) {
                    if (plans.size() >= 1024) { plans.clear(); }
                    it = plans.emplace(key, planBand(H, W, L.cin_pad, L.cout_pad / 16, B, cus)).first;
                }
                bp = it->second;
            }
            recent[recent_next] = Recent{key, bp, true};
            recent_next = (recent_next + 1) % 4;
        }
    }
    static const char* const force_th = getenv("MZ_BAND_TH"); // (experiments: a forced band height / tiles per job / waves, where the patch fits; read once)
    static const char* const force_gm = getenv("MZ_BAND_GM");
    static const char* const force_nw = getenv("MZ_BAND_NW");
    if (force_th) {
        const int TH = std::max(1, std::min(H, atoi(force_th))), PP = (TH + 2) * (W + 2), CS = PP + ((16 - PP % 32) + 32) % 32;
        if (bp.TH > 0 && size_t(L.cin_pad) * CS * sizeof(float) <= size_t(156) * 1024) { bp.TH = TH; bp.CS = CS; bp.lds = size_t(L.cin_pad) * CS * sizeof(float); }
        if (force_gm) { bp.GM = std::max(1, std::min(6, atoi(force_gm))); }
        if (force_nw) { bp.NW = atoi(force_nw) == 16 ? 16 : 8; }
    }
    if (bp.TH <= 0) { return MZ_OK; }
    MZ_LDS_ATTR(conv3x3_band, size_t(160) * 1024);
    hipLaunchKernelGGL(conv3x3_band, dim3(B, (H + bp.TH - 1) / bp.TH), dim3(64 * bp.NW), bp.lds, s, in, L.cin, L.cin_pad / 4, params + L.w_off, params + L.b_off, skip, out,
                       L.cout, L.cout_pad / 16, H, W, bp.TH, bp.CS, bp.GM);
    MZ_HIP(hipGetLastError());
    *launched = true;
    return MZ_OK;
}

int Net::launchConvAny(const ConvLayer& L, const float* in, const float* skip, float* out, int B)
{
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width;
    static const bool no_band = getenv("MZ_NO_CONV_BAND") != nullptr; // (A/B: the round-5 kernel that reads its B operand from global memory)
    if (!no_band) {
        bool launched = false;
        const int rc = launchConvBand(L, params_.p, in, skip, out, B, H, W, stream_, cu_count_, &launched);
        if (rc != MZ_OK || launched) { return rc; }
    }
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
