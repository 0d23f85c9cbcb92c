// Device bodies of the network kernels that are shared by the stand-alone kernels (net.hip) and the per-game simulation
// kernel (sim.hip): the fused residual tower and the fused heads.  gfx950, -ffp-contract=off; numerics contract in DESIGN.md §4.
#pragma once
#include "net_dev.h"
#include <type_traits>

#ifndef MZ_HPROF
#define MZ_HPROF(k) // experiment hook (sim.hip -DMZ_SIM_HPROF): time stamps inside the heads
#endif

namespace mz {

// ---------------------------------------------------------------------------------------------
// tower_fused — the whole trunk (stem conv + 2*num_blocks residual convs) of one sample in ONE workgroup:
// activations never leave LDS between layers.  Three zero-bordered LDS tiles [CMAX][CS] rotate as
// (input, temp, output/skip); each layer is the same tap-major MFMA chain as conv3x3_mfma, its epilogue
// (bias + skip + ReLU) writes straight into the interior of the next layer's padded tile.  8 wave64 per
// workgroup (2 per SIMD): wave w owns output-channel tile (w & 3) and half of the pixel tiles, so MFMA issue of
// one wave hides the LDS/global latency of its SIMD partner.  Weights stream from L2 (147 KB per layer), one
// tap ahead in registers.  Removes 12 kernel boundaries, 12 LDS re-stagings and all inter-layer HBM traffic.
// ---------------------------------------------------------------------------------------------
struct TowerArgs {
    int nlayers, cin0, C, OT; // C = hidden channels (== cout of every layer), OT = ceil(C/16)
    int in_bits;              // input planes arrive bit-packed (1 bit per point, ceil(P/32) words per channel)
    int has_stem;             // 1: layer 0 is a stem conv (cin0 -> C); 0: the input already has C channels and layer 0 starts a residual block
    unsigned w_off[48], b_off[48];
};

// Which pixels form MFMA pixel tile t.
//  * The B operand of v_mfma_f32_16x16x4_f32 is one ds_read_b32 per lane with lane = 16 * k + n reading channel plane k at the padded
//    position of pixel n.  The four planes sit 16 banks apart (plane stride % 32 == 16), so the read is conflict-free iff the 16
//    positions are distinct mod 16.  16 CONSECUTIVE pixels of a W-wide board span up to 16 + 2 * (rows crossed) padded positions, i.e.
//    always collide (measured: SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles), so tile t takes, for each residue n, the t-th pixel whose
//    padded position is n mod 16 (pixels that do not fit fill the free slots).
//  * A board of 16 m + 1 points (9x9 = 81) needs m + 1 tiles, the last one with a single real column.  That column is made the
//    top-right CORNER pixel: only 4 of its 9 taps are inside the board, the other 5 read zero padding for every channel, and
//    a k-step whose B operand is all zero leaves the accumulator as it is (the chain starts at +0 and fma(a, 0, acc) = acc), so the
//    last tile issues 4 / 9 of its MFMAs (9x9: 784 instead of 864 MFMAs per SIMD and layer).
//    (Computing that pixel on the vector ALUs instead — 5 full tiles, 150 us per launch without it — was tried twice: the 256 dependent
//    fmas + 256 LDS reads per layer cost more issue slots beside the MFMAs than the tile saves: 181 us; round 2, with the A fragments the
//    wave already holds spread over the rows by v_permlane16_swap / v_permlane32_swap (no extra weight traffic, bit-identical outputs):
//    ~14 vector instructions per k-group instead of one MFMA, tower 148 -> 178 us per simulation.  A vector instruction beside the MFMAs
//    costs 8-11 cycles of MFMA issue on this SIMD, so the corner would have to fit in 3 instructions per k-group; the chain alone has 4.)
template <int H, int W>
struct TileMap {
    static constexpr int P = H * W, PW = W + 2;
    static constexpr int PT = (P + 15) / 16;                  // MFMA pixel tiles
    static constexpr bool kCorner = (P % 16 == 1) && H > 1 && W > 1 && PT >= 2;
    static constexpr int VQ = kCorner ? W - 1 : -1;           // the corner pixel (0, W - 1), alone in the last tile
    static constexpr int PT0 = (PT + 1) / 2, PT1 = PT - PT0;  // tiles of waves 0-3 / waves 4-7
    short q[PT * 16];
    constexpr TileMap() : q{}
    {
        for (int i = 0; i < PT * 16; ++i) { q[i] = -1; }
        constexpr int full = kCorner ? PT - 1 : PT; // tiles filled by residue class
        int cnt[16] = {};
        short over[P + 1] = {};
        int nover = 0;
        for (int i = 0; i < P; ++i) {
            if (i == VQ) { continue; }
            const int r = ((i / W + 1) * PW + (i % W) + 1) & 15;
            if (cnt[r] < full) { q[cnt[r]++ * 16 + r] = short(i); } else { over[nover++] = short(i); }
        }
        for (int s = 0, k = 0; s < full * 16 && k < nover; ++s) { if (q[s] < 0) { q[s] = over[k++]; } }
        if (kCorner) { q[(PT - 1) * 16 + (((0 + 1) * PW + (W - 1) + 1) & 15)] = short(VQ); }
    }
};
template <int H, int W>
__device__ const TileMap<H, W> kTileMap{};

// taps of the 3x3 window that are inside the board for the corner pixel (0, W - 1): (dy, dx) in {0, +1} x {-1, 0}
__host__ __device__ constexpr bool cornerTapInside(int t) { return t == 3 || t == 4 || t == 6 || t == 7; }

// One conv3x3 layer for the NT pixel tiles [tile0, tile0 + NT) and oc-tile `ot` of this wave; CORNER: the last of them is the corner tile.
// a_first / have_first: this layer's tap-0 A-fragments if the previous layer already fetched them; next_wp / a_next: the NEXT layer's
// weights (nullptr: none) whose tap-0 fragments are fetched during this layer's last tap, so that the layer boundary (epilogue,
// barrier) does not end with an exposed L2 round trip; the bias values are fetched at the start of the layer for the same reason.
// weights of a tower whose working set does not fit the L2 next to the heads' (muzero_atari: 2.1 MB + 2.5 MB per XCD of 4 MB) are streamed
// with non-temporal loads so that they do not evict the heads' weights (heads 65 -> 58 us per simulation on BASELINE configs[4])
template <bool NTW>
__device__ __forceinline__ float4 loadW4(const float* p)
{
    if constexpr (NTW) {
        typedef float vf4 __attribute__((ext_vector_type(4)));
        const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}

#ifndef MZ_TPROF
#define MZ_TPROF(slot) // tools/tower_prof.hip defines it: time stamps of wave phases inside a layer
#endif

// per-lane geometry of a wave's pixel tiles, computed once per tower (the tile map is a table in constant memory: fetching it at the start of
// every layer cost ~800 cycles of exposed latency per layer)
template <int NT>
struct PixSet {
    int off[NT];  // top-left tap of the 3x3 window in the padded plane, channel (lane >> 4)
    int dst[NT];  // interior position of the pixel in a padded plane; padding columns of a tile: the plane's first spare float (never read)
    int q[NT];    // pixel index, -1: padding column of the tile
};
template <int H, int W, int NT>
__device__ __forceinline__ PixSet<NT> makePixSet(int lane, int tile0)
{
    constexpr int PW = W + 2, CS = planeStride(H, W);
    static_assert(CS > (H + 2) * (W + 2), "the plane stride leaves a spare float behind the padded plane");
    PixSet<NT> px;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        px.q[j] = kTileMap<H, W>.q[(tile0 + j) * 16 + (lane & 15)];
        const int q = px.q[j] < 0 ? 0 : px.q[j];
        px.dst[j] = px.q[j] < 0 ? (H + 2) * (W + 2) : (q / W + 1) * PW + (q % W) + 1;
        px.off[j] = (lane >> 4) * CS + (q / W) * PW + (q % W);
    }
    return px;
}

// XOUT (cluster mode of the simulation kernel, sim_cluster.h): the oc-tile's outputs also go to `xout` ([16][P] floats in global memory, tagged with
// `xsign` in the sign bit), from where the other workgroups of the game's cluster fetch them
// (the geometry of the LDS planes as template arguments: CS = plane stride, PW = row stride of the padded plane, DUMP = a spare float of every plane that lanes of
// padding columns write, P = pixels per sample in `gout` — the board kernels pass those of one padded H x W board, sim_rounds.hip those of several boards
// stacked in one plane)
template <int CS, int PW, int DUMP, int P, int CG, int NT, int CGN, bool CORNER, bool NTW = false, bool XOUT = false>
__device__ __forceinline__ void tower_layer_geo(const float* __restrict__ tin, const float* __restrict__ tskip, float* __restrict__ tout,
                                            float* __restrict__ gout, const float* __restrict__ wp, const float* __restrict__ bias, int cout, int OT,
                                            int lane, int ot, const PixSet<NT>& px, bool have_first, float (&a_first)[CG],
                                            const float* __restrict__ next_wp, float (&a_next)[CGN], float* __restrict__ xout = nullptr, unsigned xsign = 0)
{
    const int (&pixoff)[NT] = px.off;
    const int (&pixdst)[NT] = px.dst;
    const int (&pixq)[NT] = px.q;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const float4 bias4 = *reinterpret_cast<const float4*>(bias + 16 * ot + 4 * (lane >> 4)); // the 4 output channels of this lane's accumulators
    const float biasv[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
    // A-fragments in the interleaved layout of weights.cpp (w4_off): for (tap, oc-tile) CG * 64 contiguous floats
    constexpr int CG4 = CG / 4;
    const float* wl = wp + size_t(ot) * CG * 64;
    const size_t wstep = size_t(OT) * CG * 64; // per tap
    // A-fragments are double-buffered by TAP in two register sets; the loop is fully unrolled (exact s_waitcnt counts: a loop-carried
    // prefetch gets vmcnt(0)) and the fences keep the scheduler from sinking the global loads below the tap's MFMAs (it did: every
    // tap paid an exposed L2 round trip) or hoisting all nine taps' loads (256 VGPRs + spills).  Prefetching the B operand (LDS) a tap
    // ahead the same way does not work: under that register pressure the compiler re-materialises the loads next to their uses.
    float a0[CG], a1[CG];
    auto loadFrom = [&](float* a, const float* base) { // base = first float of a (tap, oc-tile) block of CG * 64 floats
#pragma unroll
        for (int c4 = 0; c4 < CG4; ++c4) {
            const float4 w = loadW4<NTW>(base + c4 * 256 + lane * 4);
            a[4 * c4] = w.x; a[4 * c4 + 1] = w.y; a[4 * c4 + 2] = w.z; a[4 * c4 + 3] = w.w;
        }
#pragma unroll
        for (int cg = 4 * CG4; cg < CG; ++cg) { a[cg] = base[CG4 * 256 + (cg - 4 * CG4) * 64 + lane]; }
        asm volatile("" ::: "memory");
    };
    auto loadA = [&](float* a, int t) { loadFrom(a, wl + size_t(t) * wstep); };
    // B operand (one ds_read_b32 per MFMA): the values of k-group (t, cg + 1) are read BEFORE the MFMAs of group (t, cg) are issued and
    // the order is pinned with sched_barrier: left to itself the scheduler puts every read right in front of its MFMA, which costs nothing
    // while the SIMD's other wave fills the pipe, but a wave that is alone on its SIMD (the other one finished its tiles: tools/tower_prof)
    // then issues one MFMA per LDS round trip (85 cycles instead of 32)
    auto bload = [&](float (&b)[NT], int t, int cg) {
        const int tapoff = (t / 3) * PW + (t % 3);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (CORNER && j == NT - 1 && !cornerTapInside(t)) { continue; } // all-zero B operand: the k-step leaves the accumulator unchanged
            b[j] = tin[pixoff[j] + cg * 4 * CS + tapoff];
        }
    };
    float bc[NT];
    auto tap = [&](const float* a, int t) { // bc = the B values of group (t, 0) on entry, of group (t + 1, 0) on exit
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            float bn[NT];
            if (cg + 1 < CG) { bload(bn, t, cg + 1); } else if (t < 8) { bload(bn, t + 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (CORNER && j == NT - 1 && !cornerTapInside(t)) { continue; }
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cg], bc[j], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) { bc[j] = bn[j]; }
        }
    };
    bload(bc, 0, 0);
    MZ_TPROF(0);
    // A narrow layer (the stem of a board-game tower: 18-20 input planes = 5 k-groups per tap) has 480 cycles of MFMAs per tap and wave, less than an L2 round
    // trip: double-buffered by tap it waits for its fragments at every tap (the stem of BASELINE configs[1]: 12.6 k cycles for 7.8 k of MFMAs).  Its nine taps
    // are 45 registers: all of them are fetched up front, one round trip for the layer.
    constexpr bool kAllTaps = CG * 9 <= 48;
    float aall[kAllTaps ? 9 : 1][CG];
    if constexpr (kAllTaps) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t == 0 && have_first) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) { aall[0][cg] = a_first[cg]; }
            } else {
                loadA(aall[t], t);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) { tap(aall[t], t); }
    } else {
    if (have_first) {
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) { a0[cg] = a_first[cg]; }
    } else {
        loadA(a0, 0);
    }
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
        loadA(a1, t + 1);
        tap(a0, t);
        loadA(a0, t + 2);
        tap(a1, t + 1);
    }
    }
    if (next_wp) { // the next layer's first tap (its block of this wave's oc-tile): in flight during the last tap, the epilogue and the barrier
        constexpr int CGN4 = CGN / 4;
        const float* nb = next_wp + size_t(ot) * CGN * 64;
#pragma unroll
        for (int c4 = 0; c4 < CGN4; ++c4) {
            const float4 w = loadW4<NTW>(nb + c4 * 256 + lane * 4);
            a_next[4 * c4] = w.x; a_next[4 * c4 + 1] = w.y; a_next[4 * c4 + 2] = w.z; a_next[4 * c4 + 3] = w.w;
        }
#pragma unroll
        for (int cg = 4 * CGN4; cg < CGN; ++cg) { a_next[cg] = nb[CGN4 * 256 + (cg - 4 * CGN4) * 64 + lane]; }
        asm volatile("" ::: "memory");
    }
    // the residual inputs of the epilogue are read while the last tap's MFMAs run (unconditionally, from a readable tile: no branches)
    const bool has_skip = tskip != nullptr;
    const float* sk = has_skip ? tskip : tin;
    const int ocb = 16 * ot + 4 * (lane >> 4); // the 4 output channels of this lane's accumulators: ocb .. ocb + 3
    float skv[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { skv[j][r] = sk[(ocb + r) * CS + pixdst[j]]; }
    }
    asm volatile("" ::: "memory");
    if constexpr (kAllTaps) { tap(aall[8], 8); } else { tap(a0, 8); }
    MZ_TPROF(1);
    if (gout) { // stand-alone launch, last layer: NCHW to HBM
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int q = pixq[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[j][r] + biasv[r];
                v = v + (has_skip ? skv[j][r] : 0.0f); // without a skip: + 0 only turns -0 into +0, which the ReLU does anyway
                v = v > 0.0f ? v : 0.0f;
                if (q >= 0 && ocb + r < cout) { __builtin_nontemporal_store(v, &gout[(ocb + r) * P + q]); }
            }
        }
    } else { // into the next layer's tile: straight-line ds_write_b32 with immediate offsets; lanes of padding columns write the plane's
             // spare float, channels beyond cout (networks narrower than an oc-tile) the spare float of the lane's first channel
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float* dstp = tout + ocb * CS + pixdst[j];
            float* dump = tout + ocb * CS + DUMP;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[j][r] + biasv[r];
                v = v + (has_skip ? skv[j][r] : 0.0f);
                v = v > 0.0f ? v : 0.0f;
                float* d = (ocb + r < cout) ? dstp + r * CS : dump;
                *d = v;
                if constexpr (XOUT) { // v >= +0 (ReLU): its sign bit carries the exchange's phase (sim_cluster.h clExchange)
                    if (pixq[j] >= 0 && ocb + r < cout) { xout[(4 * (lane >> 4) + r) * P + pixq[j]] = __uint_as_float(__float_as_uint(v) | xsign); }
                }
            }
        }
    }
    MZ_TPROF(2);
}

template <int H, int W, int CG, int NT, int CGN, bool CORNER, bool NTW = false, bool XOUT = false>
__device__ __forceinline__ void tower_layer(const float* __restrict__ tin, const float* __restrict__ tskip, float* __restrict__ tout,
                                            float* __restrict__ gout, const float* __restrict__ wp, const float* __restrict__ bias, int cout, int OT,
                                            int lane, int ot, const PixSet<NT>& px, bool have_first, float (&a_first)[CG],
                                            const float* __restrict__ next_wp, float (&a_next)[CGN], float* __restrict__ xout = nullptr, unsigned xsign = 0)
{
    tower_layer_geo<planeStride(H, W), W + 2, (H + 2) * (W + 2), H * W, CG, NT, CGN, CORNER, NTW, XOUT>(tin, tskip, tout, gout, wp, bias, cout, OT, lane, ot, px, have_first,
                                                                                                       a_first, next_wp, a_next, xout, xsign);
}

// the layer sequence of one wave: oc-tile `ot` x the NT pixel tiles from `tile0` (CORNER: the last of them is the corner tile); T0 = the
// temporary (holds the stem's input on entry), T1 = x.  Every wave of the workgroup passes the same number of barriers (towerIdle).
template <int CS, int PW, int DUMP, int P, int CIN0_PAD, int CPAD, int NT, bool CORNER, bool NTW = false>
__device__ __forceinline__ void towerRunPx(const float* __restrict__ params, const TowerArgs& ta, float* __restrict__ T0, float* __restrict__ T1,
                                           float* __restrict__ gout, int lane, int ot, const PixSet<NT>& px)
{
    // tap-0 A-fragments of the next layer travel from layer to layer in registers
    float aS[CIN0_PAD / 4], aA[CPAD / 4], aB[CPAD / 4];
    bool have = false;
    if (ta.has_stem) { // stem: T0 -> T1
        const float* nw = ta.nlayers > 1 ? params + ta.w_off[1] : nullptr;
        tower_layer_geo<CS, PW, DUMP, P, CIN0_PAD / 4, NT, CPAD / 4, CORNER, NTW>(T0, nullptr, T1, ta.nlayers == 1 ? gout : nullptr, params + ta.w_off[0], params + ta.b_off[0],
                                                             ta.C, ta.OT, lane, ot, px, false, aS, nw, aA);
        have = nw != nullptr;
        __syncthreads();
        MZ_TPROF(3);
    }
    float *x = T1, *tmp = T0;
#pragma unroll 1
    for (int l = ta.has_stem; l < ta.nlayers; ++l) { // residual blocks: tmp = relu(conv1(x)); x = relu(conv2(tmp) + x) — one code copy for both convs
        const bool second = ((l - ta.has_stem) & 1) != 0, last = l + 1 == ta.nlayers;
        tower_layer_geo<CS, PW, DUMP, P, CPAD / 4, NT, CPAD / 4, CORNER, NTW>(second ? tmp : x, second ? x : nullptr, second ? x : tmp, last ? gout : nullptr,
                                                         params + ta.w_off[l], params + ta.b_off[l], ta.C, ta.OT, lane, ot, px, have, aA,
                                                         last ? nullptr : params + ta.w_off[l + 1], aB);
#pragma unroll
        for (int cg = 0; cg < CPAD / 4; ++cg) { aA[cg] = aB[cg]; }
        have = !last;
        __syncthreads();
        MZ_TPROF(3);
    }
}

template <int H, int W, int CIN0_PAD, int CPAD, int NT, bool CORNER, bool NTW = false>
__device__ __forceinline__ void towerRun(const float* __restrict__ params, const TowerArgs& ta, float* __restrict__ T0, float* __restrict__ T1,
                                         float* __restrict__ gout, int lane, int ot, int tile0)
{
    const PixSet<NT> px = makePixSet<H, W, NT>(lane, tile0);
    towerRunPx<planeStride(H, W), W + 2, (H + 2) * (W + 2), H * W, CIN0_PAD, CPAD, NT, CORNER, NTW>(params, ta, T0, T1, gout, lane, ot, px);
}

// waves without tiles (oc-tiles beyond the network's width, boards of a single pixel tile) only keep the barrier count
__device__ __forceinline__ void towerIdle(const TowerArgs& ta)
{
    for (int l = 0; l < ta.nlayers; ++l) { __syncthreads(); }
}

// the body of tower_fused for sample `b`, run by all 512 threads of a workgroup (tid 0..511); `tiles` = 3 x [CMAX][CS] floats of LDS
// out == nullptr: the last layer's activations stay in LDS; the returned pointer is that tile ([C][CS] padded planes)
template <int H, int W, int CIN0_PAD, int CPAD, bool NTW = false>
// hidden_src != nullptr (MuZero dynamics, ref muzero_network.py:32): the input is cat(hidden_src[C][P], one-hot plane of `action`;
// action_planes > 1: that many planes, plane `action` all ones)
// instead of sample b of `in` (a pass / out-of-board action gives an all-zero plane, ref go.cpp:310-315)
__device__ __forceinline__ float* towerBody(const float* __restrict__ in, const float* __restrict__ params, const TowerArgs& ta, float* __restrict__ out,
                                            int b, int tid, float* __restrict__ tiles, const float* __restrict__ hidden_src = nullptr, int action = -1,
                                            int action_planes = 1)
{
    constexpr int P = H * W, PW = W + 2, CS = planeStride(H, W);
    constexpr int CMAX = CIN0_PAD > CPAD ? CIN0_PAD : CPAD;
    const int lane = tid & 63, wave = tid >> 6;
    float* T0 = tiles;
    float* T1 = tiles + CMAX * CS;
    // zero both tiles (borders and padding channels stay zero for the whole kernel), then the sample's planes into T0
    static_assert((kTowerTiles * CMAX * CS) % 4 == 0, "16-byte zero fill");
    for (int i = tid; i < kTowerTiles * CMAX * CS / 4; i += 512) { reinterpret_cast<float4*>(tiles)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    __syncthreads();
    float* Tin = ta.has_stem ? T0 : T1; // without a stem the input IS the first block's x
    if (hidden_src) {
        const int CH = ta.cin0 - (action_planes > 1 ? action_planes : 1);
        for (int i = tid; i < CH * P; i += 512) {
            const int c = i / P, p = i - c * P;
            Tin[c * CS + (p / W + 1) * PW + (p % W) + 1] = hidden_src[i];
        }
        if (action_planes > 1) { // Atari: one plane per action, the chosen action's plane all ones (ref atari.cpp:124-130)
            const int CH0 = ta.cin0 - action_planes;
            if (action >= 0 && action < action_planes) {
                for (int p = tid; p < P; p += 512) { Tin[(CH0 + action) * CS + (p / W + 1) * PW + (p % W) + 1] = 1.0f; }
            }
        } else if (tid == 0 && action >= 0 && action < P) { Tin[CH * CS + (action / W + 1) * PW + (action % W) + 1] = 1.0f; }
    } else if (ta.in_bits) {
        constexpr int W32 = (P + 31) / 32;
        const unsigned* bits = reinterpret_cast<const unsigned*>(in) + size_t(b) * ta.cin0 * W32;
        for (int i = tid; i < ta.cin0 * P; i += 512) {
            const int c = i / P, p = i - c * P;
            Tin[c * CS + (p / W + 1) * PW + (p % W) + 1] = ((bits[c * W32 + (p >> 5)] >> (p & 31)) & 1u) ? 1.0f : 0.0f;
        }
    } else {
        const float* src = in + size_t(b) * ta.cin0 * P;
        for (int i = tid; i < ta.cin0 * P; i += 512) {
            const int c = i / P, p = i - c * P;
            Tin[c * CS + (p / W + 1) * PW + (p % W) + 1] = src[i];
        }
    }
    __syncthreads();
    float* gout = out ? out + size_t(b) * ta.C * P : nullptr;
    // wave -> (oc-tile, pixel tiles): waves 0-3 take tiles [0, PT0), waves 4-7 tiles [PT0, PT) (the last one may be the corner tile)
    using TM = TileMap<H, W>;
    const int ot = wave & 3, half = wave >> 2;
    if (ot < ta.OT && half == 0) {
        towerRun<H, W, CIN0_PAD, CPAD, TM::PT0, (TM::kCorner && TM::PT1 == 0), NTW>(params, ta, T0, T1, gout, lane, ot, 0);
    } else if (ot < ta.OT && TM::PT1 > 0) {
        if constexpr (TM::PT1 > 0) {
            // The two waves of a SIMD share its MFMA pipe.  The wave with FEWER pixel tiles has fewer independent accumulator chains (a dependent MFMA issues every
            // 57 cycles, the pipe takes one every 32): it cannot fill the pipe alone, its partner can.  Left to the arbiter the partner finished first and this wave
            // ran its last ~35 MFMAs of every layer alone at the dependent rate (tools/tower_prof: 9x9, waves 0-3 done at 25.2 k cycles, waves 4-7 at 26.3 k, the pipe's
            // 784 MFMAs are 25.1 k).  With the higher priority it is served whenever it can issue, and the partner fills everything else.  (Measured: the order
            // flips — waves 4-7 done at 24.4 k, waves 0-3 at 25.2 k on average over the 13 layers — but the later of the two is where it was: 0.2-0.5 % per move.)
            if constexpr (TM::PT1 < TM::PT0 || TM::kCorner) { __builtin_amdgcn_s_setprio(2); }
            towerRun<H, W, CIN0_PAD, CPAD, TM::PT1, TM::kCorner, NTW>(params, ta, T0, T1, gout, lane, ot, TM::PT0);
            if constexpr (TM::PT1 < TM::PT0 || TM::kCorner) { __builtin_amdgcn_s_setprio(0); }
        }
    } else {
        towerIdle(ta);
    }
    float* x = T1;
    return x;
}


// ---------------------------------------------------------------------------------------------
// fused heads (+ MuZero hidden-state rescale)
// ---------------------------------------------------------------------------------------------
struct HeadParams {
    const float *pconv_w, *pconv_b, *pfc_wT, *pfc_b, *vconv_w, *vconv_b, *vfc1_wT, *vfc1_b, *vfc2_w, *vfc2_b;
    int C, P, A, PC, VH;
};

// acc = fmaf(x[i * xs], w[i * ws], acc) for i = 0 .. n-1 IN ORDER (one f32 chain), with the weights of CH steps loaded
// ahead of the CH dependent fmas: the chain is latency-bound, and with the load issued next to each fma every step paid an L2 trip
// The weights of these chains are always in global memory; a pointer that was loaded from a parameter block is a GENERIC pointer to the compiler,
// its loads FLAT loads, and those count on lgkmcnt as well: every wait for an LDS operand (x) would wait for all weight loads in flight.
// GW = true makes them global loads (the muzero_atari heads; the board-game heads keep the pointers they always had: their code is unchanged).
template <bool GW, class T>
struct WPtr { typedef const T* type; };
template <class T>
struct WPtr<true, T> { typedef __attribute__((address_space(1))) const T* type; };
// XL: the x operand is known to lie in LDS — read with ds_read instead of through the generic pointer.  A FLAT load counts on vmcnt AND lgkmcnt and completes out
// of order with respect to global loads, so the wait for an x value loaded that way is a wait for every weight load in flight: the prefetch is void (the 722-step
// policy chains of a 19x19 board: 47 us of heads).
template <bool XL, class T>
struct XPtr { typedef const T* type; };
template <class T>
struct XPtr<true, T> { typedef __attribute__((address_space(3))) const T* type; };
template <int CH, bool GW = false, bool XL = false>
__device__ __forceinline__ void dotChainPart(float& acc, int& i0, const float* __restrict__ x_, int xs, const float* __restrict__ w_, size_t ws, int n)
{
    typename WPtr<GW, float>::type w = (typename WPtr<GW, float>::type)w_;
    typename XPtr<XL, float>::type x = (typename XPtr<XL, float>::type)x_;
    for (; i0 + CH <= n; i0 += CH) {
        float wv[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) { wv[k] = w[size_t(i0 + k) * ws]; }
#pragma unroll
        for (int k = 0; k < CH; ++k) { acc = __builtin_fmaf(x[(i0 + k) * xs], wv[k], acc); }
    }
}
template <int CH, bool GW = false, bool XL = false>
__device__ __forceinline__ float dotChain(const float* __restrict__ x, int xs, const float* __restrict__ w, size_t ws, int n)
{
    float acc = 0.0f;
    int i0 = 0;
    dotChainPart<CH, GW, XL>(acc, i0, x, xs, w, ws, n);
    if (CH > 16) { dotChainPart<16, GW, XL>(acc, i0, x, xs, w, ws, n); } // the tail of a deep prefetch in shallower groups, not one load at a time
    if (CH > 4) { dotChainPart<4, GW, XL>(acc, i0, x, xs, w, ws, n); }
    for (; i0 < n; ++i0) { acc = __builtin_fmaf(((typename XPtr<XL, float>::type)x)[i0 * xs], ((typename WPtr<GW, float>::type)w)[size_t(i0) * ws], acc); }
    return acc;
}

// K independent chains per thread (each one the same ordered f32 chain as dotChain): the weights of CH steps of all K chains are in flight
// together and the K dependent fma sequences interleave, so a latency-bound thread with several outputs finishes them in the time of one
template <int CH, int K, bool GW = false, bool XL = false>
__device__ __forceinline__ void dotChainK(const float* const (&x_)[K], int xs, const float* const (&w_)[K], size_t ws, int n, float (&acc)[K])
{
    typename WPtr<GW, float>::type w[K];
    typename XPtr<XL, float>::type x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { acc[k] = 0.0f; w[k] = (typename WPtr<GW, float>::type)w_[k]; x[k] = (typename XPtr<XL, float>::type)x_[k]; }
    int i0 = 0;
    for (; i0 + CH <= n; i0 += CH) {
        float wv[K][CH];
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int c = 0; c < CH; ++c) { wv[k][c] = w[k][size_t(i0 + c) * ws]; }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int k = 0; k < K; ++k) { acc[k] = __builtin_fmaf(x[k][(i0 + c) * xs], wv[k][c], acc[k]); }
        }
    }
    for (; i0 < n; ++i0) {
#pragma unroll
        for (int k = 0; k < K; ++k) { acc[k] = __builtin_fmaf(x[k][i0 * xs], w[k][size_t(i0) * ws], acc[k]); }
    }
}

// Four ADJACENT outputs o .. o + 3 of a fully connected layer with weights wT[i][o] (row stride `ws` floats): each of the four is its own
// ordered f32 chain over i, and every step fetches the four weights with ONE 16-byte load — a wave has at most 63 loads in flight, so four
// times the bytes per load is what a bandwidth-starved GEMV needs (one sample per CU: the weights of the 601-bin heads stream from L2).
// nvalid < 4: the tail of the layer (outputs beyond it are computed from clamped addresses and dropped by the caller).
template <int CH, bool GW = false>
__device__ __forceinline__ void dotChain4(const float* __restrict__ x, const float* __restrict__ wT, size_t ws, int o, int nout, int n, float (&acc)[4])
{
    typedef float vf4u __attribute__((ext_vector_type(4), aligned(4)));
    const int oc = o + 3 < nout ? o : (nout >= 4 ? nout - 4 : 0); // clamped so that the 16 bytes stay inside the row
    const int sh = o - oc;                                         // outputs of this thread start at component `sh` of the clamped load
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int i0 = 0;
    auto part = [&](auto chc) {
        constexpr int C2 = decltype(chc)::value;
        for (; i0 + C2 <= n; i0 += C2) {
            vf4u wv[C2];
#pragma unroll
            for (int k = 0; k < C2; ++k) { wv[k] = *(typename WPtr<GW, vf4u>::type)(wT + size_t(i0 + k) * ws + oc); }
#pragma unroll
            for (int k = 0; k < C2; ++k) {
                const float xv = x[i0 + k];
                a0 = __builtin_fmaf(xv, wv[k].x, a0);
                a1 = __builtin_fmaf(xv, wv[k].y, a1);
                a2 = __builtin_fmaf(xv, wv[k].z, a2);
                a3 = __builtin_fmaf(xv, wv[k].w, a3);
            }
        }
    };
    part(std::integral_constant<int, CH>{});
    part(std::integral_constant<int, 4>{});
    part(std::integral_constant<int, 1>{});
    const float r[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc[k] = (k + sh < 4) ? r[(k + sh) & 3] : 0.0f; } // component k + sh of the clamped load is output o + k
}

// s = ((x[0] + x[1]) + x[2]) + ... in index order, by ONE wave: lane l holds elements [l * VPL, (l + 1) * VPL) in registers and the running
// sum is handed from lane to lane.  The same n - 1 dependent adds as a scalar loop, but without an LDS round trip per
// element (one lane reading x[i] and adding, 601 times, cost 33 us per sum: the 601-bin heads have four such sums).
template <int VPL>
__device__ __forceinline__ float orderedSumWaveT(const float* x, int n, int vpl, int lane)
{
    float v[VPL]; // slots beyond the lane's elements hold +0: adding +0 never changes a sum that is not -0, and these sums never are
#pragma unroll
    for (int k = 0; k < VPL; ++k) { const int i = lane * vpl + k; v[k] = (k < vpl && i < n) ? x[i] : 0.0f; }
    // Systolic: in every step each lane adds its elements to what its left neighbour held after the previous step (DPP wave_shr:1, lane 0 reads +0).
    // Lane 0 is right after step 0 and stays right (same inputs every step), so lane l is right from step l on: after `lanes` steps the last lane
    // holds the sum of all elements, added in index order.  The hand-over costs no instruction (the shift is folded into the first add of the
    // step); handing the sum over with v_readlane cost a VALU -> SGPR -> VALU round trip per lane (3.9 us per 601-bin sum, 2.3 us this way).
    float a = 0.0f;
    const int lanes = (n + vpl - 1) / vpl;
    for (int l = 0; l < lanes; ++l) {
        a = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
#pragma unroll
        for (int k = 0; k < VPL; ++k) { a = a + v[k]; }
    }
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), lanes - 1));
}
__device__ __forceinline__ float orderedSumWave(const float* x, int n, int lane)
{
    const int vpl = (n + 63) / 64;
    if (vpl <= 4) { return orderedSumWaveT<4>(x, n, vpl, lane); }
    if (vpl <= 10) { return orderedSumWaveT<10>(x, n, vpl, lane); } // 601 bins
    if (vpl <= 16) { return orderedSumWaveT<16>(x, n, vpl, lane); }
    float s = 0.0f; // not reached by the supported head sizes: plain loop
    for (int i = 0; i < n; ++i) { s += x[i]; }
    return s;
}

// the body of heads_kernel for sample `b`, run by NT threads (a multiple of 64, >= 128); `sm` = (C*P + PC*P + P + VH + A + 16) floats of LDS
// With xlds and without scale_hidden it passes exactly TWO workgroup barriers (after the conv1x1, after the FCs): the simulation kernel runs the second half
// of the Go leaf beside it on waves that have no share of the heads (sim.hip simLeafRest / go_body.h goLeafBody PART 2), and those waves pass the same two.
// BIGA (boards of more than 128 actions, 13x13 / 19x19 Go: the policy FC is a 722-step chain per logit at 19x19): the weights of the long chains 64 steps ahead, and the
// value FC1 on the threads the policy FC leaves free (two hidden units per thread, interleaved) instead of behind it on the same threads.  Same chains, same bits.
// FP ("fast pointers", simulation kernels only: activations and scratch are LDS, weights global memory): every chain reads its weights with GLOBAL loads and its x
// operand with DS reads instead of through the generic pointers (flat loads: their completion is counted on both counters, so that the wait for an LDS operand waits
// for every weight in flight).  BIGA implies it.
template <bool BIGA = false, bool FP = BIGA>
__device__ __forceinline__ void headsBody(const float* __restrict__ x, const HeadParams& hp, float* __restrict__ policy, float* __restrict__ logit,
                                          float* __restrict__ value, float* __restrict__ hidden_dst, const int* __restrict__ dst_idx, int scale_hidden,
                                          int b, int tid, int NT, float* __restrict__ sm, const float* __restrict__ xlds = nullptr, int xcs = 0,
                                          int xpw = 0)
{
    // xlds != nullptr: the activations are already in LDS as padded planes (channel stride xcs, row stride xpw, 1-pixel border): no copy,
    // `sm` then only holds the scratch (PC*P + P + VH + A + 16 floats); with scale_hidden the tile is rescaled in place
    const int C = hp.C, P = hp.P, A = hp.A, PC = hp.PC, VH = hp.VH;
    float* xs = sm;                // [C*P]
    float* pf = xlds ? sm : xs + C * P; // [PC*P]
    float* vf = pf + PC * P;       // [P]
    float* h1 = vf + P;            // [VH]
    float* lg = h1 + VH;           // [A] logits, then exp values
    float* red = lg + A;           // [16] reduction scratch
    const int lane = tid & 63, wave = tid >> 6, NW = NT >> 6;
    if (!xlds) {
        const float* src = x + size_t(b) * C * P;
        for (int i = tid; i < C * P; i += NT) { xs[i] = src[i]; }
        __syncthreads();
    }
    const float* xrd = xlds ? xlds : xs;
    const int xstride = xlds ? xcs : P;

    if (scale_hidden) { // min/max are order-free; (h - min) / scale is one IEEE op each
        float* xw = xlds ? const_cast<float*>(xlds) : xs;
        auto xidx = [&](int i) {
            if (!xlds || xpw == 0) { return i; } // (xpw == 0: dense planes — activations too large for the LDS, read where they lie in global memory)
            const int c = i / P, p = i - c * P;
            return c * xcs + (p / (xpw - 2) + 1) * xpw + p % (xpw - 2) + 1;
        };
        float mn = 3.4e38f, mx = -3.4e38f;
        for (int i = tid; i < C * P; i += NT) { float v = xw[xidx(i)]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        for (int o = 32; o > 0; o >>= 1) {
            float m2 = __shfl_xor(mn, o), x2 = __shfl_xor(mx, o);
            mn = m2 < mn ? m2 : mn;
            mx = x2 > mx ? x2 : mx;
        }
        if (lane == 0) { red[wave] = mn; red[8 + wave] = mx; }
        __syncthreads();
        mn = red[0]; mx = red[8];
        for (int w = 1; w < NW; ++w) { mn = red[w] < mn ? red[w] : mn; mx = red[8 + w] > mx ? red[8 + w] : mx; }
        float scale = mx - mn;
        if (scale < 1e-5f) { scale += 1e-5f; }
        float* hd = hidden_dst + size_t(dst_idx ? dst_idx[b] : b) * C * P;
        for (int i = tid; i < C * P; i += NT) {
            const int k = xidx(i);
            float v = (xw[k] - mn) / scale;
            xw[k] = v;
            hd[i] = v;
        }
        __syncthreads();
    }

    MZ_HPROF(5);
    // conv1x1 + folded BN + ReLU: PC policy planes and 1 value plane, one output element per thread
    for (int i = tid; i < (PC + 1) * P; i += NT) {
        const int j = i / P, p = i - j * P;
        const float* w = (j < PC) ? hp.pconv_w + j * C : hp.vconv_w;
        const int xo = (xlds && xpw != 0) ? (p / (xpw - 2) + 1) * xpw + p % (xpw - 2) + 1 : p;
        const float acc = (FP && xlds && xpw != 0) ? dotChain<16, true, true>(xrd + xo, xstride, w, 1, C) : dotChain<16>(xrd + xo, xstride, w, 1, C);
        float v = acc + ((j < PC) ? hp.pconv_b[j] : hp.vconv_b[0]);
        v = v > 0.0f ? v : 0.0f;
        if (j < PC) { pf[i] = v; } else { vf[p] = v; }
    }
    __syncthreads();
    MZ_HPROF(6);

    // policy FC (one logit per thread, waves 0..) and value FC1 (one hidden unit per thread, on other waves when there are enough)
    for (int a = tid; a < A; a += NT) {
        const float v = (BIGA ? dotChain<64, true, true>(pf, 1, hp.pfc_wT + a, A, PC * P) : FP ? dotChain<16, true, true>(pf, 1, hp.pfc_wT + a, A, PC * P) : dotChain<16>(pf, 1, hp.pfc_wT + a, A, PC * P)) + hp.pfc_b[a];
        lg[a] = v;
        logit[size_t(b) * A + a] = v;
    }
    if (BIGA && NT == 512 && A <= 384) {
        if (tid >= 384) {
            for (int o = tid - 384; o < VH; o += 256) { // hidden units o and o + 128: two independent chains of P steps, interleaved
                const int o2 = o + 128 < VH ? o + 128 : o;
                const float* const xs2[2] = {vf, vf};
                const float* const ws2[2] = {hp.vfc1_wT + o, hp.vfc1_wT + o2};
                float acc2[2];
                dotChainK<32, 2, true, true>(xs2, 1, ws2, VH, P, acc2);
                const float v0 = acc2[0] + hp.vfc1_b[o], v1 = acc2[1] + hp.vfc1_b[o2];
                h1[o] = v0 > 0.0f ? v0 : 0.0f;
                if (o + 128 < VH) { h1[o2] = v1 > 0.0f ? v1 : 0.0f; }
            }
        }
    } else {
        const int vo = (NT >= 256 && A <= 128) ? 128 : 0; // first thread of the value FC1 group
        for (int o = (tid - vo + NT) % NT; o < VH; o += NT) {
            const float v = (FP ? dotChain<16, true, true>(vf, 1, hp.vfc1_wT + o, VH, P) : dotChain<16>(vf, 1, hp.vfc1_wT + o, VH, P)) + hp.vfc1_b[o];
            h1[o] = v > 0.0f ? v : 0.0f;
        }
    }
    __syncthreads();
    MZ_HPROF(7);

    // value FC2 + tanh: one sequential chain (wave 1, lane 0) while wave 0 does the softmax
    if (tid == 64) {
        const float acc = BIGA ? dotChain<32, true, true>(h1, 1, hp.vfc2_w, 1, VH) : FP ? dotChain<16, true, true>(h1, 1, hp.vfc2_w, 1, VH) : dotChain<16>(h1, 1, hp.vfc2_w, 1, VH);
        value[b] = mz_tanhf(acc + hp.vfc2_b[0]);
    }
    if (wave == 0) {
        float m = -3.4e38f;
        for (int a = lane; a < A; a += 64) { m = lg[a] > m ? lg[a] : m; }
        for (int o = 32; o > 0; o >>= 1) { float m2 = __shfl_xor(m, o); m = m2 > m ? m2 : m; }
        for (int a = lane; a < A; a += 64) { lg[a] = mz_expf(lg[a] - m); }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const float s = orderedSumWave(lg, A, lane); // index-order sum (the lanes hand it on: 82 dependent adds of one lane each behind an LDS read cost 1 us of the tail's 2.2)
        for (int a = lane; a < A; a += 64) { policy[size_t(b) * A + a] = lg[a] / s; }
        MZ_HPROF(8);
    }
}


} // namespace mz
