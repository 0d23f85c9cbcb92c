// Opt-in 16-bit-input residual tower (`mz_nn_precision=bf16x3`): the same fused tower as net_body.h — one sample per workgroup, all layers
// without leaving LDS — on `v_mfma_f32_16x16x32_bf16` with SPLIT operands.  Every f32 value v is carried as hi = bf16(v), lo = bf16(v - hi)
// (16 mantissa bits together) and a product x * w is computed as x_hi * w_hi + x_hi * w_lo + x_lo * w_hi with f32 accumulation inside the
// MFMA (the x_lo * w_lo term, 2^-16 of the product, is dropped): three MFMAs at 16x the f32-input rate.  Network outputs stay within the
// north star's 1e-3 of the f32 path (tests/test_gpu_bf16.py), but the summation order inside a K = 32 MFMA cannot be mirrored by the CPU
// oracle, so records are NOT bit-identical to the reference in this mode: the default (and the headline benchmark) stays the f32 tower.
//
// Data layout (what makes the B operand one conflict-free ds_read_b128): activations live in LDS channel-innermost as bf16, hi and lo in
// separate buffers, a lane's B fragment = 8 consecutive channels ("chunk") of one padded board position.  A 16-lane service group of
// ds_read_b128 reads 16 different positions of one chunk; positions of a pixel tile are distinct mod 16 (TileMap), so chunk c of position p is
// stored at 16-byte slot (p mod 16) of row (p / 16) * 8 + c: the group's 16 reads fall on the 16 slots of 256-byte rows — no bank conflict.
#pragma once
#include "net_body.h"

namespace mz {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct TowerArgsBf16 {
    int nlayers, cin0, C, OT;
    unsigned w_off[48]; // per layer: offset of its fragments in 16-byte units: [tap][oc-tile][k-block][hi, lo][lane] x 8 bf16
    unsigned b_off[48]; // per layer: bias offset (floats) in the f32 parameter blob
};

template <int H, int W>
struct Bf16Geom {
    static constexpr int PW = W + 2, NPOS = (H + 2) * (W + 2), NPOS16 = (NPOS + 16) / 16 * 16; // at least one spare position (the dump slot)
    static constexpr int kBufBytes = NPOS16 * 128;  // one buffer (hi or lo) of 64 channels
    static constexpr int kDump = NPOS16 - 1;
};
__device__ __forceinline__ int actByte(int pos, int chunk) { return (((pos >> 4) * 8 + chunk) * 16 + (pos & 15)) * 16; }

template <int NT>
struct PixSetBf16 {
    int src[NT];  // padded position of the window's top-left tap
    int dst[NT];  // padded position of the pixel itself (padding columns of a tile: the dump position)
    int q[NT];
};

// One conv3x3 layer (64 -> 64, or the stem with KB = 1 and no lo input) for oc-tile `ot` and NT pixel tiles.
// out_f32 != nullptr: last layer, f32 padded planes [C][CS] for the heads instead of the bf16 pair.
// A fragments travel kAhead steps ahead of their MFMAs in a register ring (a (tap, k-block) step is ~150 cycles of MFMA issue, an L2 round trip
// several times that); the first kAhead steps of the NEXT layer (next_wf, NKB k-blocks per tap) are fetched before this layer's epilogue and
// handed over in registers (ring_hi / ring_lo), so that a layer does not start with an exposed round trip.
constexpr int kBf16Ahead = 4;
template <int H, int W, int KB, int NT, bool CORNER, bool HAS_LO, int NKB>
__device__ __forceinline__ void tower_layer_bf16(const char* __restrict__ in_hi, const char* __restrict__ in_lo, const char* __restrict__ skip_hi,
                                                 const char* __restrict__ skip_lo, char* __restrict__ out_hi, char* __restrict__ out_lo,
                                                 float* __restrict__ out_f32, float* __restrict__ gout, const uint4* __restrict__ wf, const float* __restrict__ bias,
                                                 int OT, int lane, int ot, const PixSetBf16<NT>& px, bool have_first, uint4 (&ring_hi)[kBf16Ahead],
                                                 uint4 (&ring_lo)[kBf16Ahead], const uint4* __restrict__ next_wf)
{
    using G = Bf16Geom<H, W>;
    constexpr int PW = G::PW, CS = planeStride(H, W), P = H * W, STEPS = 9 * KB, D = kBf16Ahead;
    const int kg = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const float4 bias4 = *reinterpret_cast<const float4*>(bias + 16 * ot + 4 * kg);
    const float biasv[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
    const uint4* wme = wf + size_t(ot) * KB * 128 + lane;   // step s = (tap, kb): wme[(tap * OT * KB + kb) * 128]
    auto loadStep = [&](int s, uint4& hi, uint4& lo) {
        const uint4* p = wme + (size_t(s / KB) * OT * KB + (s % KB)) * 128;
        hi = p[0];
        lo = p[64];
    };
    if (!have_first) {
#pragma unroll
        for (int s = 0; s < D; ++s) { loadStep(s, ring_hi[s], ring_lo[s]); }
    }
    // B fragments one step ahead of their MFMAs (the order is pinned: left alone the scheduler puts every LDS read right in front of its
    // MFMA, and two waves per SIMD then stall on the LDS together)
    auto bload = [&](int s, uint4 (&bh)[NT], uint4 (&bl)[NT]) {
        const int t = s / KB, kb = s % KB;
        const int tapoff = (t / 3) * PW + (t % 3);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (CORNER && j == NT - 1 && !cornerTapInside(t)) { continue; }
            const int off = actByte(px.src[j] + tapoff, kb * 4 + kg);
            bh[j] = *reinterpret_cast<const uint4*>(in_hi + off);
            if constexpr (HAS_LO) { bl[j] = *reinterpret_cast<const uint4*>(in_lo + off); }
        }
    };
    uint4 b_hi[NT], b_lo[NT];
    bload(0, b_hi, b_lo);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int t = s / KB;
        uint4 n_hi[NT], n_lo[NT];
        if (s + 1 < STEPS) { bload(s + 1, n_hi, n_lo); }
        const bf16x8 ah = __builtin_bit_cast(bf16x8, ring_hi[s % D]), al = __builtin_bit_cast(bf16x8, ring_lo[s % D]);
        if (s + D < STEPS) { loadStep(s + D, ring_hi[s % D], ring_lo[s % D]); } // this slot's fragments are in `ah` / `al` now
        else if (next_wf) { // the next layer's step s + D - STEPS
            const int ns = s + D - STEPS;
            const uint4* p = next_wf + size_t(ot) * NKB * 128 + lane + (size_t(ns / NKB) * OT * NKB + (ns % NKB)) * 128;
            ring_hi[s % D] = p[0];
            ring_lo[s % D] = p[64];
        }
        __builtin_amdgcn_sched_barrier(0);
        // product-major: consecutive MFMAs write different accumulators (three on one accumulator back to back wait for each other)
        if constexpr (HAS_LO) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (CORNER && j == NT - 1 && !cornerTapInside(t)) { continue; } // all-zero B operand: nothing to add
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, b_lo[j]), acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (CORNER && j == NT - 1 && !cornerTapInside(t)) { continue; }
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, b_hi[j]), acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (CORNER && j == NT - 1 && !cornerTapInside(t)) { continue; }
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, b_hi[j]), acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) { b_hi[j] = n_hi[j]; if constexpr (HAS_LO) { b_lo[j] = n_lo[j]; } }
    }
    if (next_wf) { // the ring now holds the next layer's steps in slot order (STEPS + i) % D for step i: rotate to slot i
        uint4 th[D], tl[D];
#pragma unroll
        for (int i = 0; i < D; ++i) { th[i] = ring_hi[(STEPS + i) % D]; tl[i] = ring_lo[(STEPS + i) % D]; }
#pragma unroll
        for (int i = 0; i < D; ++i) { ring_hi[i] = th[i]; ring_lo[i] = tl[i]; }
    }
    // epilogue: + bias (+ skip), ReLU; D layout of the 16x16 MFMA: lane (n = lane & 15, kg) holds output channels 16 * ot + 4 * kg + r at pixel n
    const int ocb = 16 * ot + 4 * kg;
    const int chunk = ocb >> 3, half = (ocb >> 2) & 1;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        float v[4];
        float skv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (skip_hi) {
            const int off = actByte(px.dst[j], chunk) + half * 8;
            const bf16x4 sh = *reinterpret_cast<const bf16x4*>(skip_hi + off), sl = *reinterpret_cast<const bf16x4*>(skip_lo + off);
#pragma unroll
            for (int r = 0; r < 4; ++r) { skv[r] = static_cast<float>(sh[r]) + static_cast<float>(sl[r]); }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = acc[j][r] + biasv[r];
            x = x + skv[r];
            v[r] = x > 0.0f ? x : 0.0f;
        }
        if (gout) { // stand-alone launch, last layer: NCHW f32 to HBM
            const int q = px.q[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) { if (q >= 0) { __builtin_nontemporal_store(v[r], &gout[(ocb + r) * P + q]); } }
        } else if (out_f32) { // simulation kernel, last layer: f32 padded planes for the heads
            const int q = px.q[j];
            const int d = q < 0 ? (H + 2) * (W + 2) : (q / W + 1) * PW + (q % W) + 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) { out_f32[(ocb + r) * CS + d] = v[r]; }
        } else {
            bf16x4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hi[r] = static_cast<__bf16>(v[r]);
                lo[r] = static_cast<__bf16>(v[r] - static_cast<float>(hi[r]));
            }
            const int off = actByte(px.dst[j], chunk) + half * 8;
            *reinterpret_cast<bf16x4*>(out_hi + off) = hi;
            *reinterpret_cast<bf16x4*>(out_lo + off) = lo;
        }
    }
}

template <int H, int W, int NT, bool CORNER>
__device__ __forceinline__ void towerRunBf16(const uint4* __restrict__ wfrag, const float* __restrict__ params, const TowerArgsBf16& ta, char* bufs,
                                             float* __restrict__ out_f32, float* __restrict__ gout, int lane, int ot, int tile0)
{
    using G = Bf16Geom<H, W>;
    constexpr int PW = G::PW;
    PixSetBf16<NT> px;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        px.q[j] = kTileMap<H, W>.q[(tile0 + j) * 16 + (lane & 15)];
        const int q = px.q[j] < 0 ? 0 : px.q[j];
        px.src[j] = (q / W) * PW + (q % W);
        px.dst[j] = px.q[j] < 0 ? G::kDump : (q / W + 1) * PW + (q % W) + 1;
    }
    char* xh = bufs;                      // x: the blocks' input / output (and the skip)
    char* xl = bufs + G::kBufBytes;
    char* th = bufs + 2 * G::kBufBytes;   // the temporary; holds the stem's input planes (hi only: they are 0 / 1) on entry
    char* tl = bufs + 3 * G::kBufBytes;
    uint4 ring_hi[kBf16Ahead], ring_lo[kBf16Ahead];
    // stem: t -> x
    tower_layer_bf16<H, W, 1, NT, CORNER, false, 2>(th, nullptr, nullptr, nullptr, xh, xl, nullptr, nullptr, wfrag + ta.w_off[0], params + ta.b_off[0], ta.OT, lane,
                                                   ot, px, false, ring_hi, ring_lo, wfrag + ta.w_off[1]);
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l < ta.nlayers; ++l) { // residual blocks: t = relu(conv1(x)); x = relu(conv2(t) + x)
        const bool second = ((l - 1) & 1) != 0, last = l + 1 == ta.nlayers;
        tower_layer_bf16<H, W, 2, NT, CORNER, true, 2>(second ? th : xh, second ? tl : xl, second ? xh : nullptr, second ? xl : nullptr, second ? xh : th,
                                                       second ? xl : tl, last ? out_f32 : nullptr, last ? gout : nullptr, wfrag + ta.w_off[l],
                                                       params + ta.b_off[l], ta.OT, lane, ot, px, true, ring_hi, ring_lo,
                                                       last ? nullptr : wfrag + ta.w_off[l + 1]);
        __syncthreads();
    }
}

// LDS of the bf16 tower: four activation buffers (+ the f32 planes of the last layer when they stay in LDS)
template <int H, int W>
constexpr int towerBf16LdsBytes(bool f32_out) { return 4 * Bf16Geom<H, W>::kBufBytes + (f32_out ? 64 * planeStride(H, W) * 4 : 0); }

// the body for sample `b` (all 512 threads): input = bit-packed planes (cin0 <= 32 channels); C must be 64.
// out != nullptr: last activations as f32 NCHW to HBM; else they stay in LDS as f32 padded planes and the returned pointer is that tile.
template <int H, int W>
__device__ __forceinline__ float* towerBodyBf16(const unsigned* __restrict__ in_bits, const uint4* __restrict__ wfrag, const float* __restrict__ params,
                                                const TowerArgsBf16& ta, float* __restrict__ out, int b, int tid, char* __restrict__ bufs)
{
    using G = Bf16Geom<H, W>;
    constexpr int P = H * W, PW = G::PW, W32 = (P + 31) / 32;
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * G::kBufBytes / 16; i += 512) { reinterpret_cast<uint4*>(bufs)[i] = uint4{0, 0, 0, 0}; }
    float* xf = reinterpret_cast<float*>(bufs + 4 * G::kBufBytes);
    if (!out) { for (int i = tid; i < 64 * planeStride(H, W); i += 512) { xf[i] = 0.0f; } }
    __syncthreads();
    {
        char* th = bufs + 2 * G::kBufBytes;
        const unsigned* bits = in_bits + size_t(b) * ta.cin0 * W32;
        for (int i = tid; i < ta.cin0 * P; i += 512) {
            const int c = i / P, p = i - c * P;
            if ((bits[c * W32 + (p >> 5)] >> (p & 31)) & 1u) {
                const int pos = (p / W + 1) * PW + (p % W) + 1;
                *reinterpret_cast<unsigned short*>(th + actByte(pos, c >> 3) + (c & 7) * 2) = 0x3F80; // bf16 1.0
            }
        }
    }
    __syncthreads();
    float* gout = out ? out + size_t(b) * ta.C * P : nullptr;
    using TM = TileMap<H, W>;
    const int ot = wave & 3, half = wave >> 2;
    if (half == 0) {
        towerRunBf16<H, W, TM::PT0, (TM::kCorner && TM::PT1 == 0)>(wfrag, params, ta, bufs, out ? nullptr : xf, gout, lane, ot, 0);
    } else if (TM::PT1 > 0) {
        if constexpr (TM::PT1 > 0) { towerRunBf16<H, W, TM::PT1, TM::kCorner>(wfrag, params, ta, bufs, out ? nullptr : xf, gout, lane, ot, TM::PT0); }
    } else {
        for (int l = 0; l < ta.nlayers; ++l) { __syncthreads(); }
    }
    return xf;
}

} // namespace mz
