// The residual tower for shapes whose two f32 activation tiles do not fit the 160 KB of LDS next to what the simulation kernel keeps there:
// wide towers (128 / 256 hidden channels: the reference's default network is 1 block x 256 channels, config/configuration.cpp:70-72) and large
// boards (13x13, 19x19 Go: go_unit.h:11).  Same arithmetic as net_body.h (per output one k-ordered fmaf chain, tap-major, channel
// ascending = the chain of v_mfma_f32_16x16x4_f32 steps), another data flow:
//   * ONE LDS tile [C][CS] holds the current layer's input (zero-padded planes).  Shapes whose waves hold all their accumulators at once (WideGeo::kSinglePass:
//     every instance so far) run a layer as MFMAs -> workgroup barrier -> epilogue, and the epilogue writes the outputs IN PLACE over the tile the layer has
//     just read; the block input x also goes to a per-workgroup block in global memory (L2-resident), where the second conv of the block finds its skip
//     values — read back by the very lane that wrote them.  (Shapes with several passes per wave stage every layer through global memory: wideRestage.)
//     The bias, the skip values (where they are few) and the next layer's first A unit (where the registers allow) are fetched ahead of the barrier.
//   * wave w of the 8 owns NOT = OT / WO adjacent output-channel tiles (WO = min(OT, 8) wave columns) x the pixel tiles of its pixel group
//     (8 / WO groups), in passes of at most 12 accumulator tiles; pixel tiles are 16 CONSECUTIVE pixels (coalesced 64-byte runs to global memory;
//     the B operand's bank conflicts were measured not to matter).
//   * the tap loop is a real loop (a 256-channel layer fully unrolled would be 110 KB of code): A-fragments travel in units of up to four dwordx4
//     chunks (16 k-steps) per oc-tile, double-buffered in two register sets; weights in the `wq` layout (weights.cpp: [tap][oc-tile][chunk][lane][4],
//     input channels padded to 16, which IS the fused tower's w4 layout when the layer's channels are a multiple of 16).
// gfx950, -ffp-contract=off.
#pragma once
#include "net_body.h"

namespace mz {

template <int H_, int W_, int C_>
struct WideGeo {
    static constexpr int H = H_, W = W_, C = C_;
    static constexpr int P = H * W, PW = W + 2, PP = (H + 2) * (W + 2);
    // plane stride: the two-tile tower's (stride % 32 == 16) while the tile leaves the simulation kernel its 40 KB; a tight one (multiple of 4, one spare float) beyond
    static constexpr int CS = (C * planeStride(H, W) * 4 <= 120 * 1024) ? planeStride(H, W) : ((PP + 1 + 3) & ~3);
    static constexpr int PT = (P + 15) / 16;
    static constexpr bool kCorner = (P % 16 == 1) && PT >= 2; // the last tile holds the single pixel (H-1, W-1): 4 of its 9 taps are inside the board
    static constexpr int OT = C / 16;
    static constexpr int WO = OT < 8 ? OT : 8, NOT = OT / WO, WP = 8 / WO;
    static constexpr int NTMAX = 12 / NOT;
    static_assert(C % 16 == 0 && (OT == 1 || OT == 2 || OT == 4 || OT == 8 || OT == 16), "wide tower: 16, 32, 64, 128 or 256 hidden channels");
    static constexpr int groupTiles(int wp) { return PT / WP + (wp < PT % WP ? 1 : 0); }
    static constexpr int groupStart(int wp) { int s = 0; for (int i = 0; i < wp; ++i) { s += groupTiles(i); } return s; }
    static constexpr int passes(int wp) { return (groupTiles(wp) + NTMAX - 1) / NTMAX; }
    static constexpr int passTiles(int wp, int ps) { const int n = groupTiles(wp), k = passes(wp); return k == 0 ? 0 : n / k + (ps < n % k ? 1 : 0); }
    static constexpr int passStart(int wp, int ps) { int s = groupStart(wp); for (int i = 0; i < ps; ++i) { s += passTiles(wp, i); } return s; }
    static constexpr bool singlePass() { for (int wp = 0; wp < WP; ++wp) { if (passes(wp) > 1) { return false; } } return true; }
    static constexpr bool kSinglePass = singlePass(); // every wave holds all its accumulators at once: the layers' outputs go in place into the tile
    static constexpr int maxPassTiles() { int m = 0; for (int wp = 0; wp < WP; ++wp) { for (int ps = 0; ps < passes(wp); ++ps) { m = passTiles(wp, ps) > m ? passTiles(wp, ps) : m; } } return m; }
    // the next layer's first A unit is fetched ahead only where its 16 registers do not push the kernel into spills (measured: 19x19 x 64 with 12 tiles per wave and
    // the two-oc-tile waves of 256 channels are at the 256-VGPR limit without it)
    static constexpr bool kPrefetchNext = NOT == 1 && maxPassTiles() <= 8;
};
template <int H, int W, int C>
constexpr size_t wideTileFloats(int cin0q) { return size_t(cin0q > C ? cin0q : C) * WideGeo<H, W, C>::CS; }

// taps of the 3x3 window that are inside the board for the corner pixel (H - 1, W - 1): (dy, dx) in {-1, 0} x {-1, 0}
__host__ __device__ constexpr bool lastCornerTapInside(int t) { return t == 0 || t == 1 || t == 3 || t == 4; }

template <int I, int N, class F>
__device__ __forceinline__ void wideStaticFor(F& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wideStaticFor<I + 1, N>(f);
    }
}

// One pass of one conv3x3 layer: NOT oc-tiles from `ot0` x the NT pixel tiles from `tile0`.  CQ = dwordx4 chunks (16 input channels) per (tap, oc-tile).
// tout != nullptr (shapes whose waves run ONE pass): the outputs go IN PLACE into the tile the layer read — behind a workgroup barrier that every wave of the
// workgroup passes once all MFMAs have been issued (waves without a pass: wideConv) — and to gout only where a later layer reads them as skip values (gout may
// be nullptr); no trip through global memory between two layers.  tout == nullptr: the outputs go to gout, the caller stages them back (wideRestage).
// wq_next / apre (in-place shapes): the NEXT layer's first unit of A-fragments (CQN chunks per tap there) is fetched into `apre` before this layer's barrier, and this
// layer takes its own first unit from `apre` when the previous one left it there (have_pre): a layer then does not start with an exposed trip to the L2.
template <class G, int CQ, int NT, int NOT, bool CORNER, int CQN = CQ>
__device__ __forceinline__ void wideLayerPass(const float* __restrict__ tin, const float* gskip, float* gout, const float* __restrict__ wq,
                                              const float* __restrict__ bias, int cout, int lane, int ot0, int tile0, float* tout, const float* __restrict__ wq_next,
                                              float (&apre)[NOT][NOT >= 2 ? 8 : 16], bool have_pre)
{
    constexpr int CS = G::CS, PW = G::PW, P = G::P, W = G::W, OT = G::OT;
    // A-fragments travel in units of UC dwordx4 chunks per oc-tile (4 chunks = 16 k-steps; 2 where a wave owns two oc-tiles: the two register sets of a
    // 256-channel layer would otherwise be 64 VGPRs and the kernel spills), UPT units per tap
    constexpr int UC = NOT >= 2 ? 2 : 4, UPT = (CQ + UC - 1) / UC;
    f32x4 acc[NOT][NT];
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    }
    int pixoff[NT]; // channel (lane >> 4) at the top-left tap of the pixel's window
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int q = 16 * (tile0 + j) + (lane & 15);
        if (q >= P) { q = 0; } // padding column of the last tile: any readable position, the column is dropped
        pixoff[j] = (lane >> 4) * CS + (q / W) * PW + (q % W);
    }
    float a[2][NOT][4 * UC];
    auto loadUnit = [&](float (&buf)[NOT][4 * UC], int t, int u) { // the A-fragments of unit u of tap t: (chunks of the unit) x NOT dwordx4 loads
        const int nch = u < UPT - 1 ? UC : CQ - UC * (UPT - 1);
#pragma unroll
        for (int i = 0; i < NOT; ++i) {
            const float* base = wq + (size_t(t * OT + ot0 + i) * CQ + UC * u) * 256 + lane * 4;
#pragma unroll
            for (int c = 0; c < UC; ++c) {
                if (c < nch) {
                    const float4 w = *reinterpret_cast<const float4*>(base + c * 256);
                    buf[i][4 * c] = w.x; buf[i][4 * c + 1] = w.y; buf[i][4 * c + 2] = w.z; buf[i][4 * c + 3] = w.w;
                }
            }
        }
        asm volatile("" ::: "memory");
    };
    float bc[NT];
    auto setTap = [&](int (&p)[NT], int t) {
        const int tapoff = (t / 3) * PW + (t % 3);
#pragma unroll
        for (int j = 0; j < NT; ++j) { p[j] = pixoff[j] + tapoff; }
    };
    // the k-steps of unit U of a tap (compile-time U): B values of step (cg + 1) are read before the MFMAs of step cg (net_body.h: a wave alone on its SIMD
    // otherwise issues one MFMA per LDS round trip); the first step of the NEXT unit / tap is read during the last step of this one
    auto unitSteps = [&](auto uc, const float (&buf)[NOT][4 * UC], const int (&p)[NT], const int (&pn)[NT], bool corner_in, bool corner_in_next, bool last_tap) {
        constexpr int U = decltype(uc)::value;
        constexpr int nsteps = (U < UPT - 1 ? UC : CQ - UC * (UPT - 1)) * 4;
#pragma unroll
        for (int s = 0; s < nsteps; ++s) {
            const int cg = 4 * UC * U + s;
            float bn[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) { bn[j] = bc[j]; }
            if (s + 1 < nsteps || U + 1 < UPT) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (CORNER && j == NT - 1 && !corner_in) { continue; }
                    bn[j] = tin[p[j] + (cg + 1) * 4 * CS];
                }
            } else if (!last_tap) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (CORNER && j == NT - 1 && !corner_in_next) { continue; }
                    bn[j] = tin[pn[j]];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (CORNER && j == NT - 1 && !corner_in) { continue; } // all-zero B operand: the k-step leaves the accumulator unchanged
#pragma unroll
                for (int i = 0; i < NOT; ++i) { acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(buf[i][s], bc[j], acc[i][j], 0, 0, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) { bc[j] = bn[j]; }
        }
    };
    // one tap: its UPT units, the buffer of unit 0 = a[PAR]; every unit fetches its successor (the next tap's first unit at the end; the last tap fetches
    // tap 8 again: harmless, and the loop body stays uniform)
    auto tapBody = [&](auto par, int t, const int (&p)[NT], const int (&pn)[NT]) {
        constexpr int PAR = decltype(par)::value;
        const bool cin_ = !CORNER || lastCornerTapInside(t), cin_next = !CORNER || lastCornerTapInside(t + 1), last_tap = t == 8;
        const int tn = t < 8 ? t + 1 : 8;
        auto oneUnit = [&](auto uc) {
            constexpr int U = decltype(uc)::value;
            constexpr int cur = (PAR + U) & 1;
            if constexpr (U + 1 < UPT) { loadUnit(a[cur ^ 1], t, U + 1); } else { loadUnit(a[cur ^ 1], tn, 0); }
            if constexpr (!CORNER) { unitSteps(uc, a[cur], p, pn, true, true, last_tap); }
            else if (cin_) { unitSteps(uc, a[cur], p, pn, true, cin_next, last_tap); }
            else { unitSteps(uc, a[cur], p, pn, false, cin_next, last_tap); }
        };
        wideStaticFor<0, UPT>(oneUnit);
    };
    if (have_pre) {
#pragma unroll
        for (int i = 0; i < NOT; ++i) {
#pragma unroll
            for (int c = 0; c < 4 * UC; ++c) { a[0][i][c] = apre[i][c]; }
        }
    } else {
        loadUnit(a[0], 0, 0);
    }
    // the folded-BN bias of this wave's output channels: fetched here, used in the epilogue (there it was an exposed trip to the L2 per layer)
    // (where a wave owns two oc-tiles — 256 channels, at the register limit — it is fetched after the k-loop as before)
    float bv[NOT][4];
    auto loadBias = [&]() {
#pragma unroll
        for (int i = 0; i < NOT; ++i) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * (ot0 + i) + 4 * (lane >> 4));
            bv[i][0] = b4.x; bv[i][1] = b4.y; bv[i][2] = b4.z; bv[i][3] = b4.w;
        }
    };
    if constexpr (NOT == 1) { loadBias(); }
    int p0[NT], p1[NT];
    setTap(p0, 0);
#pragma unroll
    for (int j = 0; j < NT; ++j) { bc[j] = tin[p0[j]]; }
    // the skip values (the block's input x, in global memory): every one is loaded before the first store — gskip may BE gout, the second conv of a block writes the
    // block's output over its input lane by lane — and, where they are few (<= 24 registers), before the last tap, so that they arrive under its MFMAs
    float sk[NOT][NT][4];
    auto loadSkips = [&]() {
#pragma unroll
        for (int i = 0; i < NOT; ++i) {
            const int ocb = 16 * (ot0 + i) + 4 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int q = 16 * (tile0 + j) + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) { sk[i][j][r] = (gskip && q < P && ocb + r < cout) ? gskip[(ocb + r) * P + q] : 0.0f; }
            }
        }
        asm volatile("" ::: "memory");
    };
    constexpr bool kEarlySkips = NT * NOT <= 6;
    if constexpr (UPT % 2 == 0) { // the buffer parity returns to a[0] after every tap
#pragma unroll 1
        for (int t = 0; t < 8; ++t) {
            setTap(p1, t + 1);
            tapBody(std::integral_constant<int, 0>{}, t, p0, p1);
#pragma unroll
            for (int j = 0; j < NT; ++j) { p0[j] = p1[j]; }
        }
    } else { // odd: two taps per iteration
#pragma unroll 1
        for (int t = 0; t < 8; t += 2) {
            setTap(p1, t + 1);
            tapBody(std::integral_constant<int, 0>{}, t, p0, p1);
            setTap(p0, t + 2);
            tapBody(std::integral_constant<int, 1>{}, t + 1, p1, p0);
        }
    }
    if constexpr (kEarlySkips) { loadSkips(); }
    tapBody(std::integral_constant<int, 0>{}, 8, p0, p1);
    if constexpr (!kEarlySkips) { loadSkips(); }
    if constexpr (NOT != 1) { loadBias(); }
    if (wq_next) { // the next layer's first unit: in flight during the barrier and the epilogue
        constexpr int nchn = UC < CQN ? UC : CQN;
#pragma unroll
        for (int i = 0; i < NOT; ++i) {
            const float* base = wq_next + size_t(ot0 + i) * CQN * 256 + lane * 4;
#pragma unroll
            for (int c = 0; c < nchn; ++c) {
                const float4 w = *reinterpret_cast<const float4*>(base + c * 256);
                apre[i][4 * c] = w.x; apre[i][4 * c + 1] = w.y; apre[i][4 * c + 2] = w.z; apre[i][4 * c + 3] = w.w;
            }
        }
        asm volatile("" ::: "memory");
    }
    // epilogue: folded-BN bias (+ skip) + ReLU.  D layout: column = lane & 15 (pixel), rows 4 * (lane >> 4) + r.
    if (tout) { __syncthreads(); } // every wave has read its last B operand: the tile may be overwritten
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
        const int ocb = 16 * (ot0 + i) + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int q = 16 * (tile0 + j) + (lane & 15);
            const int pd = pixoff[j] - (lane >> 4) * CS + PW + 1; // the pixel's own position in a padded plane
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] + bv[i][r];
                v = v + sk[i][j][r]; // without a skip: + 0 only turns -0 into +0, which the ReLU does anyway
                v = v > 0.0f ? v : 0.0f;
                if (q < P && ocb + r < cout) {
                    if (tout) { tout[(ocb + r) * CS + pd] = v; }
                    if (gout) { gout[(ocb + r) * P + q] = v; }
                }
            }
        }
    }
}

// does any wave of the workgroup run a pass of N pixel tiles (CORNER: that ends with the corner tile)?  Only those variants are instantiated.
template <class G>
constexpr bool wideUsesNT(int n, bool corner)
{
    for (int wp = 0; wp < G::WP; ++wp) {
        for (int ps = 0; ps < G::passes(wp); ++ps) {
            if (G::passTiles(wp, ps) == n && (G::kCorner && G::passStart(wp, ps) + n == G::PT) == corner) { return true; }
        }
    }
    return false;
}
template <class G>
struct WidePre { float a[G::NOT][G::NOT >= 2 ? 8 : 16]; bool have; }; // the next layer's first unit of A-fragments, fetched ahead (wideLayerPass)

template <class G, int CQ, int CQN, int N = 1>
__device__ __forceinline__ void wideDispatchNT(int nt, bool corner, const float* __restrict__ tin, const float* gskip, float* gout, const float* __restrict__ wq,
                                               const float* __restrict__ bias, int cout, int lane, int ot0, int tile0, float* tout, const float* __restrict__ wq_next, WidePre<G>& pre)
{
    if constexpr (N <= 12) {
        if constexpr (wideUsesNT<G>(N, false)) { if (nt == N && !corner) { wideLayerPass<G, CQ, N, G::NOT, false, CQN>(tin, gskip, gout, wq, bias, cout, lane, ot0, tile0, tout, wq_next, pre.a, pre.have); return; } }
        if constexpr (wideUsesNT<G>(N, true)) { if (nt == N && corner) { wideLayerPass<G, CQ, N, G::NOT, true, CQN>(tin, gskip, gout, wq, bias, cout, lane, ot0, tile0, tout, wq_next, pre.a, pre.have); return; } }
        wideDispatchNT<G, CQ, CQN, N + 1>(nt, corner, tin, gskip, gout, wq, bias, cout, lane, ot0, tile0, tout, wq_next, pre);
    }
}

// one conv3x3 layer by the 8 waves of the workgroup: wave -> (oc-tiles, pixel group), the group's passes one after the other; one code copy per distinct pass size.
// tout == nullptr: outputs to gout, no barrier inside.  tout != nullptr (G::kSinglePass shapes): outputs in place into the tile (+ gout where given), ONE workgroup
// barrier inside (between the last MFMA and the first write); the caller passes the barrier behind the layer.
// wq_next (in-place shapes only): the weights of the layer that follows, CQN chunks per tap — its first unit is fetched ahead into `pre`
template <class G, int CQ, int CQN = CQ>
__device__ __forceinline__ void wideConv(const float* __restrict__ tin, const float* gskip, float* gout, const float* __restrict__ wq,
                                         const float* __restrict__ bias, int cout, int wave, int lane, float* tout, const float* __restrict__ wq_next, WidePre<G>& pre)
{
    wave = __builtin_amdgcn_readfirstlane(wave);
    const int wo = wave % G::WO, wp = wave / G::WO, ot0 = wo * G::NOT;
    const int np = G::passes(wp);
    if (tout && np == 0) { __syncthreads(); return; } // a wave without tiles keeps the barrier count
#pragma unroll 1
    for (int ps = 0; ps < np; ++ps) {
        const int nt = G::passTiles(wp, ps), t0 = G::passStart(wp, ps);
        wideDispatchNT<G, CQ, CQN>(nt, G::kCorner && t0 + nt == G::PT, tin, gskip, gout, wq, bias, cout, lane, ot0, t0, tout, G::kPrefetchNext ? wq_next : nullptr, pre);
    }
    pre.have = G::kPrefetchNext && wq_next != nullptr;
}

// a layer's outputs [C][P] (global, written by this workgroup before the barrier the caller has passed) into the interior of the tile's padded planes
template <class G>
__device__ __forceinline__ void wideRestage(const float* __restrict__ src, float* __restrict__ tile, int tid)
{
    constexpr int N = G::C * G::P, P = G::P, W = G::W, PW = G::PW, CS = G::CS;
    constexpr int N4 = N / 4;
    // four consecutive floats per thread and step (C * P is a multiple of 4: C is a multiple of 16); they may straddle a row or a plane
    for (int i4 = tid; i4 < N4; i4 += 512) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * i4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        int c = (4 * i4) / P, p = 4 * i4 - c * P;
        int y = p / W, x = p - y * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tile[c * CS + (y + 1) * PW + x + 1] = vv[k];
            if (++x == W) { x = 0; if (++y == G::H) { y = 0; ++c; } }
        }
    }
}

// The body of the wide tower for sample `b`, run by all 512 threads of a workgroup.  `tile` = max(C, CIN0Q) x CS floats of LDS; `gx`, `gt` = this workgroup's two
// blocks of C x P floats in global memory (x and the temporary).  The last layer's activations end in `gx` AND (to_lds) in the tile, whose pointer is returned
// (padded planes, channel stride G::CS, row stride W + 2).  Inputs as towerBody: f32 planes, bit-packed planes (ta.in_bits), or cat(hidden_src, action plane).
template <int H, int W, int CIN0Q, int C>
__device__ __forceinline__ float* wideTowerBody(const float* __restrict__ in, const float* __restrict__ params, const TowerArgs& ta, float* __restrict__ gx,
                                                float* __restrict__ gt, int b, int tid, float* __restrict__ tile, bool to_lds,
                                                const float* __restrict__ hidden_src = nullptr, int action = -1)
{
    using G = WideGeo<H, W, C>;
    constexpr int P = G::P, PW = G::PW, CS = G::CS, CMAX = CIN0Q > C ? CIN0Q : C;
    const int lane = tid & 63, wave = tid >> 6;
    static_assert((CMAX * CS) % 4 == 0, "16-byte zero fill");
    for (int i = tid; i < CMAX * CS / 4; i += 512) { reinterpret_cast<float4*>(tile)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    __syncthreads();
    if (hidden_src) { // MuZero dynamics (ref muzero_network.py:32): cat(hidden, one-hot plane of the action; pass / out of board: all zero, ref go.cpp:310-315)
        const int CH = ta.cin0 - 1;
        for (int i = tid; i < CH * P; i += 512) {
            const int c = i / P, p = i - c * P;
            tile[c * CS + (p / W + 1) * PW + (p % W) + 1] = hidden_src[i];
        }
        if (tid == 0 && action >= 0 && action < P) { tile[CH * CS + (action / W + 1) * PW + (action % W) + 1] = 1.0f; }
    } else if (ta.in_bits) {
        constexpr int W32 = (P + 31) / 32;
        const unsigned* bits = reinterpret_cast<const unsigned*>(in) + size_t(b) * ta.cin0 * W32;
        for (int i = tid; i < ta.cin0 * P; i += 512) {
            const int c = i / P, p = i - c * P;
            tile[c * CS + (p / W + 1) * PW + (p % W) + 1] = ((bits[c * W32 + (p >> 5)] >> (p & 31)) & 1u) ? 1.0f : 0.0f;
        }
    } else {
        const float* src = in + size_t(b) * ta.cin0 * P;
        for (int i = tid; i < ta.cin0 * P; i += 512) {
            const int c = i / P, p = i - c * P;
            tile[c * CS + (p / W + 1) * PW + (p % W) + 1] = src[i];
        }
    }
    __syncthreads();
    if constexpr (G::kSinglePass) {
        // in place: a layer's outputs overwrite the tile it read (behind the barrier inside wideConv); x also goes to global memory, where the second conv of
        // the block finds its skip values.  The temporary never leaves the LDS; gt is not used.
        WidePre<G> pre;
        pre.have = false;
        wideConv<G, CIN0Q / 16, C / 16>(tile, nullptr, gx, params + ta.w_off[0], params + ta.b_off[0], ta.C, wave, lane, tile, ta.nlayers > 1 ? params + ta.w_off[1] : nullptr, pre);
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l < ta.nlayers; l += 2) { // residual blocks (ref network_unit.py:14-23): t = relu(conv1(x)); x = relu(conv2(t) + x)
            wideConv<G, C / 16>(tile, nullptr, nullptr, params + ta.w_off[l], params + ta.b_off[l], ta.C, wave, lane, tile, params + ta.w_off[l + 1], pre);
            __syncthreads();
            wideConv<G, C / 16>(tile, gx, gx, params + ta.w_off[l + 1], params + ta.b_off[l + 1], ta.C, wave, lane, tile, l + 2 < ta.nlayers ? params + ta.w_off[l + 2] : nullptr, pre);
            __syncthreads();
        }
        (void)to_lds; (void)gt;
        return tile;
    }
    // stem: tile -> x
    WidePre<G> nopre;
    nopre.have = false;
    wideConv<G, CIN0Q / 16>(tile, nullptr, gx, params + ta.w_off[0], params + ta.b_off[0], ta.C, wave, lane, nullptr, nullptr, nopre);
    __syncthreads();
    if (ta.nlayers > 1 || to_lds) {
        wideRestage<G>(gx, tile, tid);
        __syncthreads();
    }
#pragma unroll 1
    for (int l = 1; l < ta.nlayers; l += 2) { // residual blocks (ref network_unit.py:14-23): t = relu(conv1(x)); x = relu(conv2(t) + x)
        wideConv<G, C / 16>(tile, nullptr, gt, params + ta.w_off[l], params + ta.b_off[l], ta.C, wave, lane, nullptr, nullptr, nopre);
        __syncthreads();
        wideRestage<G>(gt, tile, tid);
        __syncthreads();
        wideConv<G, C / 16>(tile, gx, gx, params + ta.w_off[l + 1], params + ta.b_off[l + 1], ta.C, wave, lane, nullptr, nullptr, nopre);
        __syncthreads();
        if (l + 2 < ta.nlayers || to_lds) {
            wideRestage<G>(gx, tile, tid);
            __syncthreads();
        }
    }
    return tile;
}

} // namespace mz
