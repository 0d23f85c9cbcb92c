// Cluster mode of the MuZero simulation kernel (included by sim.hip): FOUR workgroups per game.
//
// A pool of 64 games (BASELINE configs[4]: muzero_atari, 64 games per GPU) leaves 192 of the 256 CUs idle with one workgroup per game, and inside
// a game every phase is a dependent chain, so the only way to use them is to spread ONE simulation over several CUs:
//  * dynamics tower: member m of the cluster computes oc-tile m (16 output channels) of every layer on three of its waves (one pixel tile
//    each); after each layer the four members swap their 16 x P outputs through a buffer in global memory, i.e. through the L2 of the XCD
//    they share (workgroup ids congruent mod 8 are dispatched to the same XCD; the kernel checks XCC_ID and refuses to run otherwise).
//    tools/xcu_sync_bench.hip: such an exchange costs 1.1 - 1.5 us with loads that bypass the vector cache (clExchange); with agent-scope
//    release / acquire (L2 write-back + invalidate on this multi-XCD part) it costs 21 us, hence the hand-made protocol.  A layer is 3.4 us of
//    dependent MFMAs (a chain of v_mfma_f32_16x16x4_f32 issues every 57 cycles) + 1 us of exchange.  Weights are read with
//    ordinary (temporal) loads: the three waves of a member fetch the same fragments, and with streaming loads each of them went to the L2
//    (63 -> 51 us per simulation);
//  * heads: member 0 rescales + stores the hidden state and runs the policy head, member 1 the reward head, member 2 the value head.  In a pool
//    of full octets (>= 64 games) the eight games that share an XCD run each 601-bin head TOGETHER (octetHead): 16 CUs streaming 1.24 MB each got
//    0.74 TB/s out of one XCD's L2 together (27 us per simulation for the FC layers however deep each CU prefetched), so the CU of game j takes
//    column slice j % 4 of both FC layers for the four games of its half and every weight leaves the L2 twice instead of eight times;
//  * the tree phases stay on member 0 (the owner), which sends (parent slot, action) to the helpers and collects value / reward.
// The arithmetic of every output is the same chain as in the one-workgroup kernel: records are bit-identical (tests/test_gpu_worker.py
// test_atari_cluster_pools_are_equivalent, test_atari_execution_modes_are_equivalent; tests/test_gpu_baseline_nets.py against the oracle).
// Every wait is bounded: a member that times out raises the pool's error flag and the whole cluster leaves the kernel.
// BASELINE configs[4]: 154 -> 116 us per simulation (tower 93 -> 51, heads 43 -> 43 with 1/4 of the L2 traffic, tree phases 18 -> 19).
#pragma once

namespace mz {

// per-game block in global memory (32-bit words); the host clears it before every launch
constexpr int kClCmd = 0;      // [0..3] = {parent slot, action, -, sequence number} written by the owner with one 16-byte store
constexpr int kClRes = 64;     // [0,1] = (reward bits, seq), [2,3] = (value bits, seq)
constexpr int kClXcc = 96;     // [0..3] XCC_ID of the members, [4] arrivals of the placement check
constexpr int kClXbuf = 128;   // 2 x [C][P] floats
constexpr int kClMembers = 4;
constexpr int kClPollLimit = 1 << 21;
typedef unsigned clu4 __attribute__((ext_vector_type(4)));
typedef unsigned clu2 __attribute__((ext_vector_type(2)));
// LDS floats of clusterAtariHeads: the one-workgroup layout with aligned sub-buffers + the weight ring of fcStream
inline size_t clusterHeadsSmemFloats(const AtariHeadParams& hp)
{
    const int hidmax = hp.value.hidden > hp.reward.hidden ? hp.value.hidden : hp.reward.hidden;
    const int sizemax = hp.value.size > hp.reward.size ? hp.value.size : hp.reward.size;
    return atariHeadsSmemFloats(hp) + 64 + fcStreamRingFloats(hidmax, sizemax);
}
inline size_t clusterWords(int C, int P) { return size_t(kClXbuf) + 2 * size_t(C) * P + 32; }

__device__ __forceinline__ unsigned clLoadU(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a load that is served by the L2 (relaxed agent-scope atomic load = global_load_dword sc1; the compiler tracks its vmcnt like any load)
__device__ __forceinline__ float clLoadF(const float* p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void clDrain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until *p >= want (unsigned, monotonic counter); false: timed out
__device__ __forceinline__ bool clWaitGE(const unsigned* p, unsigned want)
{
    for (int i = 0; i < kClPollLimit; ++i) {
        if (clLoadU(p) >= want) { return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// Octet block: the 8 games whose workgroups share an XCD (game % 8) run each 601-bin head TOGETHER — the CU of game j computes column slice j % 4 of
// both FC layers for the four games of its half (j / 4), so the XCD's L2 delivers every weight twice per simulation instead of eight times.
// (Measured: 16 CUs streaming 1.24 MB each get 0.74 TB/s out of one XCD's L2 together, 27 us per simulation for the FC layers, however deep
// each CU prefetches.)  One block per (octet, head): [0] arrivals, then F[8][n1p], H1[8][hidp], LG[8][sizep].
constexpr int kOcHdr = 32;
__host__ __device__ inline int up4i(int v) { return (v + 3) & ~3; }
inline size_t octetWords(int n1, int hidden, int size) { return size_t(kOcHdr) + 8 * (size_t(up4i(n1)) + up4i(hidden) + up4i(size)) + 32; }

struct ClusterCtx {
    unsigned* cm;   // this game's block
    int member, C, P, OT;
    int mine0, mine1; // floats [mine0, mine1) of an exchange buffer are this member's own channels
    unsigned xseq;  // layer exchanges done so far in this launch
    int* abort_lds; // workgroup-wide abort flag
    int* err;
    unsigned* om;   // this (octet, head)'s block (helpers), nullptr: every game runs its heads alone
    int NJ, j;      // games in the octet, this game's position
    unsigned oseq;  // octet exchanges done so far in this launch
    const float* convw; // helpers: the head's conv1x1 weights + biases in LDS (nullptr: read from global memory)
};

// the helpers of an octet meet (all 512 threads of each): false = a member went missing
__device__ __forceinline__ bool octExchange(ClusterCtx& c, int tid)
{
    clDrain();
    __syncthreads();
    const unsigned k = ++c.oseq;
    if (tid == 0) {
        __hip_atomic_fetch_add(c.om, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!clWaitGE(c.om, k * unsigned(c.NJ))) { *c.abort_lds = 1; atomicExch(c.err, 94); }
    }
    __syncthreads();
    return *c.abort_lds == 0;
}

// One 601-bin head for the 8 games of a full octet, run by the head's helper of each game (512 threads): own conv1x1, then for the four games
// of this game's half (position j / 4) column slice j % 4 of FC1 and of FC2 — wave w < 4 = game 4 * (j / 4) + w, lane = unit / bins of the
// slice — then the own game's softmax expectation -> *out.  Every weight leaves the L2 twice per simulation instead of eight times.
// `stage` = LDS for the half's FC inputs (4 * (n1p + hidp) floats).
__device__ __forceinline__ bool octetHead(const DiscreteParams& d, const float* xin, int C, int P, float* f, float* lg, float* red, float* ring, float* stage,
                                          const float* convw, float* out, ClusterCtx& c, int tid)
{
    const int n1 = d.hc * P, n1p = up4i(n1), hidp = up4i(d.hidden), sizep = up4i(d.size), j = c.j;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), slice = j & 3, g0 = j & ~3; // games g0 .. g0 + 3 of the octet
    float* Fg = reinterpret_cast<float*>(c.om + kOcHdr);
    float* H1g = Fg + 8 * n1p;
    float* LGg = H1g + 8 * hidp;
    float* Fall = stage;
    float* H1all = stage + 4 * n1p;
    if (convw) { discreteConvLds<512>(convw, convw + d.hc * C, d.hc, xin, C, P, f, tid); } else { discreteConv<512>(d, xin, C, P, f, tid); }
    __syncthreads();
    MZ_HPROF(1);
    for (int i = tid; i < n1; i += 512) { Fg[j * n1p + i] = f[i]; }
    if (!octExchange(c, tid)) { return false; }
    MZ_HPROF(10);
    { // the conv outputs of my half's four games (contiguous in the block)
        const int tot = 4 * n1p;
#pragma unroll 2
        for (int i0 = 0; i0 < tot; i0 += 4 * 512) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int idx = i0 + q * 512 + tid; v[q] = clLoadF(Fg + g0 * n1p + (idx < tot ? idx : 0)); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int idx = i0 + q * 512 + tid; if (idx < tot) { Fall[idx] = v[q]; } }
        }
    }
    __syncthreads();
    MZ_HPROF(2);
    { // FC1: hidden units [u0, u0 + seg) of the half's games
        const int SL = d.hidden / 4, u0 = slice * SL, seg = SL; // octetHeadFits: hidden = 4 * SL, SL = 64, 16 or 8
        const bool wmine = wave < 4;
        const float* xv = Fall + (wave < 4 ? wave : 0) * n1p;
        const float* xk[1] = {xv};
        const int uk[1] = {lane < seg ? lane : 0};
        float acc[1];
        if (SL == 64) { fcStreamSeg<MZ_FC1_OCTET, 64>(xk, uk, wmine, d.fc1_wT, d.hidden, u0, seg, n1, ring, tid, acc); }
        else if (SL == 16) { fcStreamSeg<MZ_FC1_OCTET, 16>(xk, uk, wmine, d.fc1_wT, d.hidden, u0, seg, n1, ring, tid, acc); }
        else { fcStreamSeg<MZ_FC1_OCTET, 8>(xk, uk, wmine, d.fc1_wT, d.hidden, u0, seg, n1, ring, tid, acc); } // the small test nets
        if (wmine && lane < seg) { const float v = acc[0] + d.fc1_b[u0 + lane]; H1g[(g0 + wave) * hidp + u0 + lane] = v > 0.0f ? v : 0.0f; }
    }
    MZ_HPROF(11);
    if (!octExchange(c, tid)) { return false; }
    MZ_HPROF(12);
    {
        const int tot = 4 * hidp;
        for (int i0 = 0; i0 < tot; i0 += 4 * 512) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int idx = i0 + q * 512 + tid; v[q] = clLoadF(H1g + g0 * hidp + (idx < tot ? idx : 0)); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int idx = i0 + q * 512 + tid; if (idx < tot) { H1all[idx] = v[q]; } }
        }
    }
    __syncthreads();
    MZ_HPROF(9);
    { // FC2: bins [b0, b0 + seg) of the half's games, bins lane, lane + 64, lane + 128 of the slice per thread
        // all four slices have SL = 151 bins: the last one starts at size - SL (three bins are computed by two CUs, with identical results)
        const int SL = (d.size + 3) / 4, b0 = slice * SL + SL <= d.size ? slice * SL : d.size - SL, seg = SL;
        const bool wmine = wave < 4;
        const float* xv = H1all + (wave < 4 ? wave : 0) * hidp;
        const float* xk[3] = {xv, xv, xv};
        int uk[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { uk[k] = lane + 64 * k < seg ? lane + 64 * k : 0; }
        float acc[3];
        fcStreamSeg<MZ_FC2_OCTET, 151>(xk, uk, wmine, d.fc2_wT, d.size, b0, seg, d.hidden, ring, tid, acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int b = lane + 64 * k;
            if (wmine && b < seg) { LGg[(g0 + wave) * sizep + b0 + b] = acc[k] + d.fc2_b[b0 + b]; }
        }
    }
    if (!octExchange(c, tid)) { return false; }
    float m = -3.4e38f;
    {
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int o = tid + q * 512; v[q] = clLoadF(LGg + j * sizep + (o < d.size ? o : 0)); }
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int o = tid + q * 512; if (o < d.size) { lg[o] = v[q]; m = v[q] > m ? v[q] : m; } }
    }
    discreteTail<512>(d.size, true, m, lg, red, out, tid);
    return true;
}
// shapes the octet heads cover: FC1 slices of <= 64 units, FC2 slices of <= 192 bins (three per lane), at most 1024 bins for the last gather
__host__ __device__ inline bool octetHeadFits(const DiscreteParams& d, int P)
{
    const int n1 = d.hc * P, sl1 = d.hidden / 4, sl2 = (d.size + 3) / 4;
    return n1 % 4 == 0 && d.hidden % 4 == 0 && (sl1 == 64 || sl1 == 16 || sl1 == 8) && sl1 * 4 == d.hidden && sl2 == 151 && d.size >= sl2 && d.size <= 1024; // the instantiated slice lengths
}

// my 16 x P block of the exchange buffer for the NEXT exchange, and the tag its words carry
__device__ __forceinline__ float* clPart(const ClusterCtx& c, int ot) { return reinterpret_cast<float*>(c.cm + kClXbuf) + size_t(c.xseq & 1) * c.C * c.P + size_t(ot) * 16 * c.P; }
__device__ __forceinline__ unsigned clSign(const ClusterCtx& c) { return (((c.xseq >> 1) & 1u) ^ 1u) << 31; }

// All 512 threads of every member, after a layer: my outputs are in LDS (`tout`, padded planes) and on their way to the exchange buffer; on return
// the other members' channels are in `tout` too.  false: a member went missing (the caller leaves the kernel).
// Every word of the buffer validates itself: the layer outputs are >= +0 (ReLU), so their sign bit is free to carry the PHASE of the exchange — the
// two buffers alternate, and the phase flips each time a buffer is reused (the host clears them: phase 0 = "never written", the first use writes 1).
// A reader simply loads the words it needs past the vector cache until all of them show the expected phase: no store drain, no counter, no second
// round trip (tools/xcu_sync_bench.hip, variant 3 against 1: 1.08 instead of 1.47 us per exchange).  A member can overwrite a buffer only after it
// has read the exchange in between from ALL members, i.e. after all of them have finished reading this one.
template <int H, int W, int CPAD>
__device__ __forceinline__ bool clExchange(ClusterCtx& c, float* __restrict__ tout, int tid)
{
    constexpr int P = H * W, PW = W + 2, CS = planeStride(H, W);
    const unsigned sign = clSign(c);
    const float* xb = reinterpret_cast<const float*>(c.cm + kClXbuf) + size_t(c.xseq & 1) * c.C * c.P;
    ++c.xseq;
    __syncthreads(); // the member's own waves are done with the layer: all waves start polling together (idle waves polling for a whole layer slowed the others' loads)
    constexpr int K = (CPAD * P + 511) / 512;
    const int n = (c.OT * 16 < c.C ? c.OT * 16 : c.C) * P, mine0 = c.mine0, mine1 = c.mine1; // channels beyond C are never written
    unsigned got[K];
    bool ok = false;
    for (int polls = 0; polls < kClPollLimit; ++polls) {
        ok = true;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = tid + j * 512;
            const bool want = i < n && (i < mine0 || i >= mine1);
            got[j] = __float_as_uint(clLoadF(xb + (want ? i : 0)));
            ok = ok && (!want || (got[j] & 0x80000000u) == sign);
        }
        ok = __all(ok);
        if (ok) { break; }
        __builtin_amdgcn_s_sleep(2);
    }
    if (!ok && (tid & 63) == 0) { *c.abort_lds = 1; atomicExch(c.err, 90); }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int i = tid + j * 512;
        const bool want = i < n && (i < mine0 || i >= mine1) && i < c.C * P;
        if (want) {
            const int ch = i / P, p = i - ch * P;
            tout[ch * CS + (p / W + 1) * PW + (p % W) + 1] = __uint_as_float(got[j] & 0x7FFFFFFFu);
        }
    }
    __syncthreads();
    return *c.abort_lds == 0;
}

// the layer sequence of one wave of a member: oc-tile `ot` x pixel tile `tile` (waves without a tile: ot < 0), every wave takes part in the exchanges
template <int H, int W, int CIN0_PAD, int CPAD, bool NTW>
__device__ __forceinline__ bool towerRunCluster(const float* __restrict__ params, const TowerArgs& ta, float* __restrict__ T0, float* __restrict__ T1, int lane,
                                                int tid, int ot, int tile, ClusterCtx& c)
{
    const bool work = ot >= 0;
    const PixSet<1> px = makePixSet<H, W, 1>(lane, work ? tile : 0);
    float aS[CIN0_PAD / 4], aA[CPAD / 4], aB[CPAD / 4];
    bool have = false;
    if (ta.has_stem) {
        if (work) {
            const float* nw = ta.nlayers > 1 ? params + ta.w_off[1] : nullptr;
            tower_layer<H, W, CIN0_PAD / 4, 1, CPAD / 4, false, NTW, true>(T0, nullptr, T1, nullptr, params + ta.w_off[0], params + ta.b_off[0], ta.C, ta.OT, lane, ot, px,
                                                                           false, aS, nw, aA, clPart(c, ot), clSign(c));
            have = nw != nullptr;
        }
        if (!clExchange<H, W, CPAD>(c, T1, tid)) { return false; }
    }
    float *x = T1, *tmp = T0;
#pragma unroll 1
    for (int l = ta.has_stem; l < ta.nlayers; ++l) {
        const bool second = ((l - ta.has_stem) & 1) != 0, last = l + 1 == ta.nlayers;
        if (work) {
            tower_layer<H, W, CPAD / 4, 1, CPAD / 4, false, NTW, true>(second ? tmp : x, second ? x : nullptr, second ? x : tmp, nullptr, params + ta.w_off[l],
                                                                       params + ta.b_off[l], ta.C, ta.OT, lane, ot, px, have, aA,
                                                                       last ? nullptr : params + ta.w_off[l + 1], aB, clPart(c, ot), clSign(c));
#pragma unroll
            for (int cg = 0; cg < CPAD / 4; ++cg) { aA[cg] = aB[cg]; }
            have = !last;
        }
        MZ_HPROF(1);
        if (!clExchange<H, W, CPAD>(c, second ? x : tmp, tid)) { return false; }
        MZ_HPROF(2);
    }
    return true;
}

// dynamics trunk of one simulation on a cluster: every member fills its own copy of the input (parent hidden state + action planes), computes its
// oc-tile of every layer and ends with the complete output x in its LDS tile T1.  nullptr: aborted.
// MEMBERS = 4: member m computes oc-tile m, one pixel tile per wave; MEMBERS = 2 (the pairs of sim_pre_pair_kernel_mz): member m computes oc-tiles 2 m and
// 2 m + 1, wave w the pixel tile w % PT of oc-tile 2 m + w / PT
template <int H, int W, int CDYN_PAD, int CPAD, int MEMBERS = 4>
__device__ __forceinline__ float* towerBodyCluster(const float* __restrict__ params, const TowerArgs& ta, int tid, float* __restrict__ tiles,
                                                   const float* __restrict__ hidden_src, int action, int action_planes, ClusterCtx& c)
{
    constexpr int P = H * W, PW = W + 2, CS = planeStride(H, W);
    constexpr int CMAX = CDYN_PAD > CPAD ? CDYN_PAD : CPAD;
    using TM = TileMap<H, W>;
    static_assert(!TM::kCorner && TM::PT <= 8, "cluster mode: one wave per pixel tile, no corner tile");
    const int lane = tid & 63, wave = tid >> 6;
    float* T0 = tiles;
    float* T1 = tiles + CMAX * CS;
    for (int i = tid; i < kTowerTiles * CMAX * CS; i += 512) { tiles[i] = 0.0f; }
    __syncthreads();
    float* Tin = ta.has_stem ? T0 : T1;
    const int CH = ta.cin0 - (action_planes > 1 ? action_planes : 1);
    {
        constexpr int K = (CMAX * P + 511) / 512; // the slab slot may have been written by another CU: read it past the vector cache
        float got[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { const int i = tid + j * 512; got[j] = clLoadF(hidden_src + (i < CH * P ? i : 0)); }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = tid + j * 512;
            if (i < CH * P) { const int ch = i / P, p = i - ch * P; Tin[ch * CS + (p / W + 1) * PW + (p % W) + 1] = got[j]; }
        }
    }
    if (action_planes > 1) {
        if (action >= 0 && action < action_planes) {
            for (int p = tid; p < P; p += 512) { Tin[(CH + action) * CS + (p / W + 1) * PW + (p % W) + 1] = 1.0f; }
        }
    } else if (tid == 0 && action >= 0 && action < P) { Tin[CH * CS + (action / W + 1) * PW + (action % W) + 1] = 1.0f; }
    __syncthreads();
    if constexpr (MEMBERS == 2) {
        static_assert(2 * TM::PT <= 8, "pairs: two oc-tiles x PT pixel tiles on the 8 waves");
        const int ot = 2 * c.member + wave / TM::PT;
        const bool work = wave < 2 * TM::PT && ot < ta.OT;
        if (!towerRunCluster<H, W, CDYN_PAD, CPAD, false>(params, ta, T0, T1, lane, tid, work ? ot : -1, wave % TM::PT, c)) { return nullptr; }
    } else {
        const bool work = c.member < ta.OT && wave < TM::PT;
        if (!towerRunCluster<H, W, CDYN_PAD, CPAD, false>(params, ta, T0, T1, lane, tid, work ? c.member : -1, wave, c)) { return nullptr; }
    }
    return T1;
}

// heads of one simulation, split by member (net_atari_body.h atariHeadsBody is the one-workgroup version): 0 = rescale + slab store + policy,
// 1 = reward head (on the UNscaled state), 2 = value head; results of 1 and 2 -> the cluster block as (bits, seq) pairs
__device__ __forceinline__ void clusterAtariHeads(const float* __restrict__ xlds, int xcs, int xpw, const AtariHeadParams& hp, float* __restrict__ policy,
                                                  float* __restrict__ logit, float* __restrict__ hd, int b, int tid, float* __restrict__ sm, ClusterCtx& c,
                                                  unsigned seq, float* __restrict__ stage)
{
    if (c.member == 3) { return; }
    const int C = hp.C, P = hp.P, A = hp.A, PC = hp.PC;
    const int hcmax = hp.value.hc > hp.reward.hc ? hp.value.hc : hp.reward.hc;
    const int hidmax = hp.value.hidden > hp.reward.hidden ? hp.value.hidden : hp.reward.hidden;
    const int sizemax = hp.value.size > hp.reward.size ? hp.value.size : hp.reward.size;
    const int lane = tid & 63, wave = tid >> 6;
    MZ_HPROF(8);
    if (c.member == 0) { MZ_HPROF(0); }
    // `sm` is 16-byte aligned (the kernel pads it) and so is every sub-buffer (fcStream reads its input vector with 16-byte LDS loads)
    auto up4 = [](int v) { return (v + 3) & ~3; };
    float* xr = sm;
    float* xs = xr + up4(C * P);
    float* pf = xs + up4(C * P);
    float* lgp = pf + up4(PC * P);
    float* redp = lgp + up4(A);
    float* f = redp + 32;
    float* h1 = f + up4(hcmax * P);
    float* lg = h1 + up4(hidmax);
    float* red = lg + up4(sizemax);
    float* ring = red + 16;
    {
        const int Wb = xpw - 2;
        for (int i = tid; i < C * P; i += 512) {
            const int ch = i / P, p = i - ch * P;
            xr[i] = xlds[ch * xcs + (p / Wb + 1) * xpw + p % Wb + 1];
        }
    }
    __syncthreads();
    if (c.member != 1) { // scale_hidden_state (ref muzero_atari_network.py:189-198); member 2 keeps its own copy, member 0 also fills the slab slot
        float mn = 3.4e38f, mx = -3.4e38f;
        for (int i = tid; i < C * P; i += 512) { const float v = xr[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(mn, o), x2 = __shfl_xor(mx, o);
            mn = m2 < mn ? m2 : mn;
            mx = x2 > mx ? x2 : mx;
        }
        if (lane == 0) { redp[wave] = mn; redp[16 + wave] = mx; }
        __syncthreads();
        mn = redp[0]; mx = redp[16];
        for (int w = 1; w < 8; ++w) { mn = redp[w] < mn ? redp[w] : mn; mx = redp[16 + w] > mx ? redp[16 + w] : mx; }
        float scale = mx - mn;
        if (scale < 1e-5f) { scale += 1e-5f; }
        for (int i = tid; i < C * P; i += 512) {
            const float v = (xr[i] - mn) / scale;
            xs[i] = v;
            if (c.member == 0) { hd[i] = v; }
        }
        __syncthreads();
    }
    if (c.member != 0) {
        __shared__ float s_out;
        MZ_HPROF(0);
        if (c.om) {
            if (!octetHead(c.member == 1 ? hp.reward : hp.value, c.member == 1 ? xr : xs, C, P, f, lg, red, ring, stage, c.convw, &s_out, c, tid)) { return; }
        } else {
            discreteHead<512, true>(c.member == 1 ? hp.reward : hp.value, true, c.member == 1 ? xr : xs, C, P, f, h1, lg, red, &s_out, tid, ring);
        }
        if (tid == 0) {
            clu2 pr;
            pr.x = __float_as_uint(invertValueDev(s_out));
            pr.y = seq;
            clu2* dst = reinterpret_cast<clu2*>(c.cm + kClRes) + (c.member - 1);
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(pr) : "memory");
        }
        return;
    }
    for (int i = tid; i < PC * P; i += 512) {
        const int j = i / P, p = i - j * P;
        const float v = dotChain<16, true>(xs + p, P, hp.pconv_w + j * C, 1, C) + hp.pconv_b[j];
        pf[i] = v > 0.0f ? v : 0.0f;
    }
    __syncthreads();
    for (int a = tid; a < A; a += 512) {
        const float v = dotChain<16, true>(pf, 1, hp.pfc_wT + a, A, PC * P) + hp.pfc_b[a];
        lgp[a] = v;
        logit[size_t(b) * A + a] = v;
    }
    __syncthreads();
    if (wave == 0) {
        float m = -3.4e38f;
        for (int a = lane; a < A; a += 64) { m = lgp[a] > m ? lgp[a] : m; }
        for (int o = 32; o > 0; o >>= 1) { const float m2 = __shfl_xor(m, o); m = m2 > m ? m2 : m; }
        for (int a = lane; a < A; a += 64) { lgp[a] = mz_expf(lgp[a] - m); }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float s = 0.0f;
        for (int a = 0; a < A; ++a) { s += lgp[a]; }
        for (int a = lane; a < A; a += 64) { policy[size_t(b) * A + a] = lgp[a] / s; }
    }
    MZ_HPROF(9);
}

// grid = 4 * gpad workgroups (cooperative launch: all of them resident), workgroup id = member * gpad + game, gpad a multiple of 8
template <int H, int W, int CDYN_PAD, int CPAD>
__global__ __launch_bounds__(512) void sim_kernel_mz_cluster(const SimArgs* __restrict__ a_, int sim0, int nsims, int host_start, int games, int gpad, int pre_epoch)
{
    CSimArgs* a = (CSimArgs*)a_;
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    const int g = blockIdx.x % gpad, member = blockIdx.x / gpad, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (g >= games) { return; }
    constexpr int CM = CDYN_PAD > CPAD ? CDYN_PAD : CPAD;
    constexpr int kTileFloats = kTowerTiles * CM * planeStride(H, W);
    __shared__ int s_abort, s_cmd[4], s_cand_k; // s_cmd: parent slot, action, leaf evaluated ahead (simPreProbe)
    double* rcp_w = reinterpret_cast<double*>(tiles + kTileFloats);
    const int rcp_n = a->rcp_n;
    const int tab_n = rcp_n - 2;
    double* sqrt_w = rcp_w + rcp_n;
    float* bias_w = reinterpret_cast<float*>(sqrt_w + tab_n);
    int* spec_w = reinterpret_cast<int*>(bias_w + tab_n + (tab_n & 1));
    if (member == 0) {
        for (int i = tid; i < rcp_n; i += 512) { rcp_w[i] = a->pv.rcp_tab[i]; }
        for (int i = tid; i < tab_n; i += 512) { sqrt_w[i] = a->pv.sqrt_tab[i]; bias_w[i] = a->pv.bias_tab[i]; }
    }
    if (tid == 0) { s_abort = 0; }
    __syncthreads();
    LdsCDouble* rcp_lds = (LdsCDouble*)rcp_w;
    SpecMem spec{nullptr, (LdsCFloat*)bias_w, (LdsCDbl*)sqrt_w};
    float* head_scratch = reinterpret_cast<float*>(spec_w + kSpecWords);
    head_scratch += (-static_cast<int>(head_scratch - tiles)) & 3; // 16-byte aligned, by pointer arithmetic only: the pointer stays an LDS pointer
    const PoolView v = ldc(&a->pv);
    ClusterCtx c;
    c.cm = a->cluster + size_t(g) * a->cluster_words;
    c.member = member; c.C = a->hp.C; c.P = a->hp.P; c.OT = a->ta_dyn.OT; c.xseq = 0; c.abort_lds = &s_abort; c.err = a->err;
    c.mine0 = member * 16 * c.P; c.mine1 = c.mine0 + 16 * c.P;
    c.NJ = (games - (g & 7) + 7) / 8; c.j = g >> 3; c.oseq = 0;
    c.om = (a->cluster_oct && c.NJ == 8 && (member == 1 || member == 2)) ? a->cluster_oct + size_t((g & 7) * 2 + member - 1) * a->oct_words : nullptr;
    // placement check: the four members of a game must share an XCD (one L2), else the exchanges would read stale data
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        __hip_atomic_store(c.cm + kClXcc + member, (id & 15u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        clDrain();
        __hip_atomic_fetch_add(c.cm + kClXcc + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = clWaitGE(c.cm + kClXcc + 4, kClMembers);
        for (int m = 0; ok && m < kClMembers; ++m) { ok = clLoadU(c.cm + kClXcc + m) == (id & 15u) + 1u; }
        if (!ok) { s_abort = 1; atomicExch(a->err, 91); }
    }
    __syncthreads();
    if (s_abort) { return; }
    c.convw = nullptr;
    if (c.om) { // the head's conv1x1 weights stay in LDS for the whole launch: in the tables of the tree walk, which only the owner uses
        const AtariHeadParams hp = ldc(&a->ahp);
        const DiscreteParams& d = member == 1 ? hp.reward : hp.value;
        float* cw = reinterpret_cast<float*>(rcp_w);
        if (d.hc * (hp.C + 1) <= static_cast<int>(head_scratch - cw)) {
            for (int i = tid; i < d.hc * hp.C; i += 512) { cw[i] = d.conv_w[i]; }
            for (int i = tid; i < d.hc; i += 512) { cw[d.hc * hp.C + i] = d.conv_b[i]; }
            c.convw = cw;
        }
        __syncthreads();
    }
    unsigned long long* prof = (a->prof && member == 0) ? a->prof + size_t(g) * 8 : nullptr;
    bool gumbel_ahead = false; // (wave 0 of the owner) the Gumbel step of this simulation was computed during the previous one
    for (int s = 0; s < nsims; ++s) {
        const int slot = sim0 + s;
        const unsigned seq = unsigned(s) + 1u;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (prof) { t0 = wall_clock64(); }
        if (member == 0) {
            if (wave == 0) { simMzSelect(a, slot, s == 0 && (host_start & 1) != 0, g, lane, tiles, rcp_lds, spec, gumbel_ahead, (host_start & 4) != 0); }
            __syncthreads();
            MZ_HPROF(14);
            if (wave == 0) {
                const int len = v.path_len[g];
                const int* path = v.path + size_t(g) * v.max_depth;
                const int src = v.hslot[size_t(g) * v.cap + path[len - 2]], action = v.path_action[size_t(g) * v.max_depth + len - 1];
                // a leaf that was evaluated ahead (sim_pre_kernel_mz): its outputs are copied in, the whole cluster skips tower + heads
                const int hslot = simPreProbe(a, pre_epoch, g, slot, src, action, lane);
                const bool hit = hslot >= 0;
                if (lane == 0) {
                    s_cmd[3] = hit ? hslot : slot; // (owner only) the slab slot of this leaf's hidden state
                    clu4 cmd;
                    cmd.x = unsigned(src);
                    cmd.y = unsigned(action);
                    cmd.z = hit ? 1u : 0u;
                    cmd.w = seq;
                    s_cmd[0] = int(cmd.x); s_cmd[1] = int(cmd.y); s_cmd[2] = int(cmd.z);
                    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(c.cm + kClCmd), "v"(cmd) : "memory");
                }
            }
        } else if (tid == 0) {
            bool ok = false;
            clu4 cmd;
            for (int i = 0; i < kClPollLimit && !ok; ++i) {
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(cmd) : "v"(c.cm + kClCmd) : "memory");
                ok = cmd.w == seq;
                if (!ok) { __builtin_amdgcn_s_sleep(1); }
            }
            if (!ok) { s_abort = 1; atomicExch(a->err, 92); }
            s_cmd[0] = int(cmd.x); s_cmd[1] = int(cmd.y); s_cmd[2] = int(cmd.z);
        }
        __syncthreads();
        if (s_abort) { return; }
        // fault injection for tests/test_gpu_worker.py (MZ_NO_SPEC=4): a helper of game 0 disappears in the third simulation — every wait of the others must
        // time out, raise the pool's error flag and leave the kernel
        if ((a->no_spec & 4) && g == 0 && member == 1 && s == 2) { return; }
        if (prof) { t1 = wall_clock64(); }
        const int src = s_cmd[0], action = s_cmd[1];
        const bool hit = s_cmd[2] != 0;
        if (hit && member != 0) { continue; } // (every member of the game sees the same flag: the exchanges stay in step)
        if (!hit) {
            const float* hsrc = a->hidden + (size_t(g) * a->slots + src) * size_t(a->hp.C) * a->hp.P;
            float* xt = towerBodyCluster<H, W, CDYN_PAD, CPAD>(a->params, *(const TowerArgs*)&a->ta_dyn, tid, tiles, hsrc, action, a->action_planes, c);
            if (!xt) { return; }
            if (prof) { t2 = wall_clock64(); }
            const AtariHeadParams hp = ldc(&a->ahp);
            float* hd = a->hidden + (size_t(g) * a->slots + slot) * size_t(hp.C) * hp.P;
            clusterAtariHeads(xt, planeStride(H, W), W + 2, hp, a->policy, a->logit, hd, g, tid, head_scratch, c, seq, tiles);
            if (s_abort) { return; }
        } else if (prof) { t2 = t1; }
        if (member != 0) { __syncthreads(); continue; }
        // The candidate list and the new children only need the policy, which this workgroup has just computed: they are built while the value and reward
        // heads of the game's other workgroups are still at work (their 601-bin heads take three times as long as the policy head); the backup follows
        // when their results have arrived.
        MZ_HPROF(10);
        int cand_k = a->A; // (a leaf evaluated ahead brings its sorted candidate list along)
        if (!hit) {
            if (wave == 0) { simMzCandGather(a, g, lane, tiles, &s_cand_k); }
            __syncthreads();
            MZ_HPROF(11);
            cand_k = s_cand_k;
            if (a->cand_coop) { simCandRank(a->A, cand_k, wave, lane, tiles); }
            __syncthreads();
        }
        MZ_HPROF(12);
        if (wave == 0) {
            simMzCandExpand(a, s_cmd[3], g, lane, tiles, cand_k, false, 1, hit);
            // ... and so is the next simulation's Gumbel step (which candidate it starts from): the backup to come only adds a visit to the child on this path
            gumbel_ahead = a->use_gumbel && s + 1 < nsims && simGumbelAhead(a, slot + 1, g, lane, tiles);
        }
        if (tid == 64 && !hit) { // value and reward from the helpers (wave 1 polls while wave 0 writes the children)
            bool ok = false;
            clu4 r;
            for (int i = 0; i < kClPollLimit && !ok; ++i) {
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(c.cm + kClRes) : "memory");
                ok = r.y == seq && r.w == seq;
                if (!ok) { __builtin_amdgcn_s_sleep(1); }
            }
            if (!ok) { s_abort = 1; atomicExch(a->err, 93); }
            a->reward[g] = __uint_as_float(r.x);
            a->value[g] = __uint_as_float(r.z);
        }
        __syncthreads();
        if (s_abort) { return; }
        if (prof) { t3 = wall_clock64(); } // (MZ_SIM_PROF: "heads" ends when the helpers' results are in; the candidate list was built meanwhile)
        if (wave == 0) { simMzCandExpand(a, slot, g, lane, tiles, cand_k, false, 2); }
        __syncthreads();
        MZ_HPROF(13);
        if (prof && tid == 0) {
            const unsigned long long t4 = wall_clock64();
            prof[0] += t1 - t0; prof[1] += t2 - t1; prof[2] += t3 - t2; prof[3] += t4 - t3; prof[4] += 1;
        }
    }
}

// which XCD does workgroup i of a launch land on?  (the dispatcher deals workgroups to the XCDs round-robin: i % 8 on this part)
__global__ void xcc_probe_kernel(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = id & 15u;
    }
}
// true if workgroups whose ids differ by a multiple of `gpad` share an XCD, i.e. if the members of a cluster will (checked once per network before the
// first cluster launch; the kernel checks again and refuses to run otherwise)
static bool clusterPlacementOk(int gpad, hipStream_t s, int members = kClMembers)
{
    const int n = members * gpad;
    unsigned* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), size_t(n) * sizeof(unsigned)) != hipSuccess) { return false; }
    std::vector<unsigned> h(n, 99u);
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(n), dim3(64), 0, s, d);
    const bool copied = hipMemcpyAsync(h.data(), d, size_t(n) * sizeof(unsigned), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(d);
    if (!copied) { return false; }
    for (int i = gpad; i < n; ++i) { if (h[i] != h[i % gpad]) { return false; } }
    return true;
}

template <int H, int W, int CDYN_PAD, int CPAD>
static int launchSimMzClusterT(const SimArgs* d_args, int games, int sim0, int nsims, int host_start, size_t lds, hipStream_t s, int pre_epoch)
{
    MZ_LDS_ATTR((sim_kernel_mz_cluster<H, W, CDYN_PAD, CPAD>), lds);
    int gpad = (games + 7) / 8 * 8;
    void* params[] = {(void*)&d_args, (void*)&sim0, (void*)&nsims, (void*)&host_start, (void*)&games, (void*)&gpad, (void*)&pre_epoch};
    // all 4 * gpad workgroups must be resident at once (they wait for each other): a cooperative launch guarantees it or fails
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<void*>(sim_kernel_mz_cluster<H, W, CDYN_PAD, CPAD>), dim3(kClMembers * gpad), dim3(512), params,
                                                    static_cast<unsigned>(lds), s);
    if (e == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return 1; } // not enough free CUs: the caller falls back to one workgroup per game
    MZ_HIP(e);
    return MZ_OK;
}

} // namespace mz
