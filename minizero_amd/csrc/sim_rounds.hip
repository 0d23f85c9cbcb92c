// Gumbel rounds, batched (muzero_atari with a Gumbel root, BASELINE configs[4]): the leaves the next R simulations of every game are expected to
// reach (sim.hip sim_pre_kernel_mz explains why they can be evaluated ahead and why records cannot change) are evaluated by a PIPELINE of kernels, each shaped
// for its own bound, instead of one workgroup per leaf doing everything:
//   pre_walk_kernel   one wave per (game, r, hypothesis): the Gumbel step on a private copy + the PUCT walk -> (parent's slab slot, action) per leaf
//   pre_tower_kernel  the dynamics trunk of NL leaves per workgroup — their 6x6 boards stacked in ONE padded LDS plane, so that 4 leaves are 144 pixels = 9 exact
//                     MFMA pixel tiles (one leaf: 36 pixels on 48) and every A-fragment fetched from the L2 feeds 4-5 tiles instead of 1-2 —, then the rescale
//                     (-> slab slot) and the conv1x1 of the two 601-bin heads (-> feature-major activations of the FC layers)
//   pre_fc_kernel     FC1 (612 -> 256) and FC2 (256 -> 601) of both heads for ALL leaves as MFMA GEMMs: samples are the N dimension, so the 2.5 MB of FC weights are
//                     read once per tile of 16 (64) samples instead of once per leaf by a single CU (which bound the in-workgroup heads at one CU's L2 port);
//                     v_mfma_f32_16x16x4_f32 over ascending k IS the contract's ordered f32 chain
//   pre_tail_kernel   per leaf: the two softmax expectations (index-ordered sums, invertValue), the policy head, the candidate list in the reference's order, the key
// Every value is computed by the same chain of IEEE operations as in sim_kernel_mz / sim_pre_kernel_mz (same device functions where possible), so the entries are
// bit-identical and the simulation kernel consumes them unchanged (simPreProbe).
#include "sim_args.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>

namespace mz {

// NL boards of H x W stacked in one padded plane: a zero row above the first board and below each board (shared by neighbours) and ONE zero column per row, which is
// the left border of its row and the right border of the row above (row stride W + 1).  With W + 2 columns the 16 pixels of an MFMA tile could only fall on 12 of
// the 16 residues mod 16 (the B operand's ds_read_b32 then has two-way bank conflicts on a quarter of its lanes: 42 % conflict cycles with four leaves per
// workgroup); an odd row stride spreads the positions over all residues, and tile t takes the t-th pixel of every residue class (net_body.h TileMap).
template <int H, int W, int NL>
struct StackGeo {
    static constexpr int PW = W + 1, ROWS = 1 + NL * (H + 1), POS = ROWS * PW + 1, DUMP = POS, P = H * W, PIX = NL * P;
    static constexpr int CS = POS + 1 + ((16 - (POS + 1) % 32) + 32) % 32; // > POS (the spare float), % 32 == 16 (net_dev.h planeStride)
    static constexpr int PT = (PIX + 15) / 16, PT0 = (PT + 1) / 2, PT1 = PT - PT0;
    __host__ __device__ static constexpr int pos(int j, int p) { return (1 + j * (H + 1) + p / W) * PW + p % W + 1; }
    short q[PT * 16]; // pixel (leaf * P + point) of tile t, column n; -1: padding column
    constexpr StackGeo() : q{}
    {
        for (int i = 0; i < PT * 16; ++i) { q[i] = -1; }
        int cnt[16] = {};
        short over[PIX + 1] = {};
        int nover = 0;
        for (int i = 0; i < PIX; ++i) {
            const int r = pos(i / P, i % P) & 15;
            if (cnt[r] < PT) { q[cnt[r]++ * 16 + r] = short(i); } else { over[nover++] = short(i); }
        }
        for (int sl = 0, k = 0; sl < PT * 16 && k < nover; ++sl) { if (q[sl] < 0) { q[sl] = over[k++]; } }
    }
};
template <int H, int W, int NL>
__device__ const StackGeo<H, W, NL> kStackGeo{};

template <int H, int W, int NL, int NT>
__device__ __forceinline__ PixSet<NT> makeStackPixSet(int lane, int tile0)
{
    using G = StackGeo<H, W, NL>;
    PixSet<NT> px;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = kStackGeo<H, W, NL>.q[(tile0 + j) * 16 + (lane & 15)];
        const bool real = q >= 0;
        const int pp = real ? G::pos(q / G::P, q % G::P) : G::PW + 1;
        px.q[j] = q;
        px.dst[j] = real ? pp : G::DUMP;
        px.off[j] = (lane >> 4) * G::CS + pp - G::PW - 1;
    }
    return px;
}

// ---- 1. the walks ------------------------------------------------------------------------------------------------------------------------------------------
// ctl[leaf] = {ok, parent's slab slot, action, entry (game * slots + slot)}; leaf = blockIdx.x = (game, hypothesis, r)
__global__ __launch_bounds__(64) void pre_walk_kernel(const SimArgs* __restrict__ a_, int s0, int R, int NH, int* __restrict__ ctl_out)
{
    CSimArgs* a = (CSimArgs*)a_;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x / (R * NH), q = blockIdx.x % (R * NH), r = q % R, hyp = q / R, lane = threadIdx.x;
    const PoolView v = ldc(&a->pv);
    __shared__ int ctl[4];
    __shared__ int st_l[4 + kGumbelMaxSample];
    int* path_l = reinterpret_cast<int*>(smem); // [2 * max_depth + 2], the Gumbel step's scratch behind it
    float* gscratch = smem + ((2 * v.max_depth + 2 + 3) & ~3);
    const int slot = s0 + r + (hyp ? a->alt_base : 0);
    const int stride = 3 + kGumbelMaxSample;
    for (int i = lane; i < stride; i += 64) { st_l[i] = a->gum.state[size_t(g) * stride + i]; }
    waveSync();
    GumbelView gl = ldc(&a->gum);
    gl.state = st_l - size_t(g) * stride; // the step sorts / halves the COPY
    (void)gumbelStepBody(v, gl, s0, g, lane, gscratch);
    waveSync();
    const size_t base = size_t(g) * v.cap;
    const int fc = v.rec[base].first_child, ncand = st_l[0];
    bool ok = slot < a->slots && s0 + r < (a->alt_base ? a->alt_base : a->slots) && r < ncand && r < kGumbelMaxSample;
    if (ok) { ok = v.rec[base + fc + st_l[3 + r]].count == v.rec[base + fc + st_l[3]].count; } // still in the round of candidate 0
    ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
    int src = 0, action = 0;
    if (ok) {
        if (lane == 0) { ctl[3] = fc + st_l[3 + r]; }
        waveSync();
        const PoolView pl = simPathViewSafe(v, path_l, g);
        selectBody<false>(pl, ctl + 3 - g, g, lane, v.rcp_tab);
        waveSync();
        const int len = path_l[2 * v.max_depth];
        if (hyp == 0) {
            src = v.hslot[base + path_l[len - 2]];
            action = path_l[v.max_depth + len - 1];
        } else { // the second expected leaf (sim.hip sim_pre_kernel_mz, hypothesis 1): the first unvisited child of the expected leaf's grandparent
            ok = len >= 4;
            if (ok) {
                const NodeRec z = v.rec[base + path_l[len - 3]];
                const unsigned visn = static_cast<unsigned>(z.players) >> 16;
                ok = visn != 0xFFFFu && static_cast<int>(visn) < z.num_children;
                if (ok) {
                    src = v.hslot[base + path_l[len - 3]];
                    action = v.rec[base + z.first_child + visn].action;
                }
            }
            ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
        }
    }
    if (lane == 0) {
        int4 o;
        o.x = ok ? 1 : 0; o.y = src; o.z = action; o.w = static_cast<int>(size_t(g) * a->slots + slot);
        reinterpret_cast<int4*>(ctl_out)[blockIdx.x] = o;
    }
}

// MZ_SIM_PROF=1: shader cycles (clock64) and 100-MHz wall ticks (wall_clock64) that workgroup 0 of the trunk kernels spent between its input staging and its epilogue,
// per leaves-per-workgroup — their ratio is the shader clock under the trunks' MFMA load (printed by Net::dumpRoundsProf)
__device__ unsigned long long g_trunk_clk[3][3];

// ---- 2. the dynamics trunk of NL leaves + rescale + the heads' conv1x1 --------------------------------------------------------------------------------------
// fT: [2 heads][n1 = hc * P features][NS samples] (sample = leaf index), the B operand of FC1
template <int H, int W, int CDYN_PAD, int CPAD, int NL>
__global__ __launch_bounds__(512) void pre_tower_kernel(const SimArgs* __restrict__ a_, const int* __restrict__ ctl_in, int nleaves, float* __restrict__ fT, int NS)
{
    CSimArgs* a = (CSimArgs*)a_;
    using G = StackGeo<H, W, NL>;
    constexpr int P = G::P, CS = G::CS;
    constexpr int CM = CDYN_PAD > CPAD ? CDYN_PAD : CPAD;
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    float* T0 = tiles;           // [CM][CS] the trunk's input, then the blocks' temporary
    float* T1 = tiles + CM * CS; // [CPAD][CS] x
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int leaf0 = blockIdx.x * NL;
    __shared__ int4 s_ctl[NL];
    if (tid < NL) { s_ctl[tid] = leaf0 + tid < nleaves ? reinterpret_cast<const int4*>(ctl_in)[leaf0 + tid] : make_int4(0, 0, 0, 0); }
    static_assert((CM * CS) % 4 == 0 && (CPAD * CS) % 4 == 0, "16-byte zero fill");
    for (int i = tid; i < (CM + CPAD) * CS / 4; i += 512) { reinterpret_cast<float4*>(tiles)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    __syncthreads();
    bool any = false;
#pragma unroll
    for (int j = 0; j < NL; ++j) { any = any || s_ctl[j].x != 0; }
    if (!any) { return; }
    const AtariHeadParams hp = ldc(&a->ahp);
    const TowerArgs& ta = *(const TowerArgs*)&a->ta_dyn;
    const int C = hp.C; // hidden channels
    // input: cat(hidden[parent's slot], one plane per action with the chosen action's plane all ones) (ref muzero_atari_network.py dynamics, atari.cpp:124-130)
    {
        const int CH = ta.cin0 - a->action_planes;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            if (s_ctl[j].x == 0) { continue; }
            const int e = s_ctl[j].w, g = e / a->slots;
            const float* hsrc = a->hidden + (size_t(g) * a->slots + s_ctl[j].y) * size_t(C) * P;
            for (int i = tid; i < CH * P; i += 512) {
                const int c = i / P, p = i - c * P;
                T0[c * CS + G::pos(j, p)] = hsrc[i];
            }
            const int action = s_ctl[j].z;
            if (action >= 0 && action < a->action_planes) {
                for (int p = tid; p < P; p += 512) { T0[(CH + action) * CS + G::pos(j, p)] = 1.0f; }
            }
        }
    }
    __syncthreads();
    const bool tprof = a->prof != nullptr && blockIdx.x == 0 && tid == 0;
    unsigned long long tc0 = 0, tw0 = 0;
    if (tprof) { tc0 = clock64(); tw0 = wall_clock64(); }
    {
        const int ot = wave & 3, half = wave >> 2;
        if (ot < ta.OT && half == 0) {
            const PixSet<G::PT0> px = makeStackPixSet<H, W, NL, G::PT0>(lane, 0);
            towerRunPx<CS, G::PW, G::DUMP, P, CDYN_PAD, CPAD, G::PT0, false, false>(a->params, ta, T0, T1, nullptr, lane, ot, px);
        } else if (ot < ta.OT && G::PT1 > 0) {
            if constexpr (G::PT1 > 0) {
                if constexpr (G::PT1 < G::PT0) { __builtin_amdgcn_s_setprio(2); } // (net_body.h towerBody: the wave with fewer accumulator chains is served first)
                const PixSet<G::PT1> px = makeStackPixSet<H, W, NL, G::PT1>(lane, G::PT0);
                towerRunPx<CS, G::PW, G::DUMP, P, CDYN_PAD, CPAD, G::PT1, false, false>(a->params, ta, T0, T1, nullptr, lane, ot, px);
                if constexpr (G::PT1 < G::PT0) { __builtin_amdgcn_s_setprio(0); }
            }
        } else {
            towerIdle(ta);
        }
    }
    if (tprof) {
        constexpr int k = NL == 4 ? 2 : NL == 2 ? 1 : 0;
        g_trunk_clk[k][0] += clock64() - tc0; g_trunk_clk[k][1] += wall_clock64() - tw0; g_trunk_clk[k][2] += 1;
    }
    // x = T1; T0 is free: the leaves' dense copies of the state and the heads' conv outputs.  The NL leaves go through the steps TOGETHER, a group of 512 / NL
    // threads per leaf (one leaf after the other on all 512 threads was 4 x 4 barriers of latency-bound work: 28 us of the 300 with four leaves)
    const int n1r = hp.reward.hc * P, n1v = hp.value.hc * P, n1 = n1r > n1v ? n1r : n1v;
    constexpr int TPL = 512 / NL, WPL = TPL / 64;      // threads / waves per leaf
    float* xall = T0;                                  // [NL][C * P] the trunk's output, rescaled in place after the reward head's conv has read it
    float* redp = xall + size_t(NL) * C * P;           // [2][8] per-wave minima / maxima
    float* fall = redp + 32;                           // [NL][2][n1] conv1x1 + ReLU outputs of the two heads
    const int j = tid / TPL, t = tid - j * TPL;        // this thread's leaf and its index in the leaf's group
    const bool live = s_ctl[j].x != 0;
    float* xr = xall + size_t(j) * C * P;
    if (live) {
        for (int i = t; i < C * P; i += TPL) {
            const int c = i / P, p = i - c * P;
            xr[i] = T1[c * CS + G::pos(j, p)];
        }
    }
    __syncthreads();
    // scale_hidden_state (ref muzero_atari_network.py:189-198): exact min / max (order-free), one IEEE operation per element (net_atari_body.h atariHeadsBody);
    // the reward head's conv reads the UNscaled state: before the rescale
    {
        float mn = 3.4e38f, mx = -3.4e38f;
        if (live) { for (int i = t; i < C * P; i += TPL) { const float v = xr[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; } }
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(mn, o), x2 = __shfl_xor(mx, o);
            mn = m2 < mn ? m2 : mn;
            mx = x2 > mx ? x2 : mx;
        }
        if (lane == 0) { redp[wave] = mn; redp[16 + wave] = mx; }
        if (live) { discreteConv<TPL>(hp.reward, xr, C, P, fall + (size_t(j) * 2 + 0) * n1, t); }
    }
    __syncthreads();
    if (live) {
        float mn = redp[j * WPL], mx = redp[16 + j * WPL];
#pragma unroll
        for (int w = 1; w < WPL; ++w) { mn = redp[j * WPL + w] < mn ? redp[j * WPL + w] : mn; mx = redp[16 + j * WPL + w] > mx ? redp[16 + j * WPL + w] : mx; }
        float scale = mx - mn;
        if (scale < 1e-5f) { scale += 1e-5f; }
        float* hd = a->hidden + size_t(s_ctl[j].w) * size_t(C) * P;
        for (int i = t; i < C * P; i += TPL) {
            const float v = (xr[i] - mn) / scale;
            xr[i] = v;
            hd[i] = v;
        }
    }
    __syncthreads();
    if (live) { discreteConv<TPL>(hp.value, xr, C, P, fall + (size_t(j) * 2 + 1) * n1, t); } // the value head's conv on the rescaled state
    __syncthreads();
    // feature-major stores: NL adjacent samples per feature
    for (int i = tid; i < 2 * n1; i += 512) {
        const int h = i / n1, f = i - h * n1;
        if (f >= (h == 0 ? n1r : n1v)) { continue; }
        float* dst = fT + (size_t(h) * n1 + f) * NS + leaf0;
#pragma unroll
        for (int jj = 0; jj < NL; ++jj) { if (s_ctl[jj].x != 0) { dst[jj] = fall[(size_t(jj) * 2 + h) * n1 + f]; } }
    }
}

// ---- 3. a fully connected layer of both heads for all samples as an MFMA GEMM ---------------------------------------------------------------------------------
// out[o][n] = sum over i ASCENDING of W[i][o] * in[i][n] (+ bias, ReLU): W = wT[nin][nout] (the heads' transposed linear weights), in = [nin][NS] feature-major.
// grid = (sample tiles / NSB, oc-tiles / 4, 2 heads), 4 waves: wave w owns oc-tile 4 * blockIdx.y + w x NSB sample tiles (NSB independent accumulator chains that
// share the A fragment).  One v_mfma_f32_16x16x4_f32 per 4 inputs: k ascending inside the instruction, steps ascending = the contract's chain.  The operands of CH
// steps are fetched a chunk ahead (a dependent chain issues an MFMA every 57 cycles: CH * 57 cycles cover an L2 round trip).
// FEAT_OUT: FC1 — bias + ReLU, output feature-major [nout][NS] (the next layer's B operand); else FC2 — bias, output sample-major [NS][ldo] (the tail reads a sample's bins)
struct FcHead { const float *wT, *bias; const float* in; float* out; int nin, nout; }; // (the two heads differ in their hidden width: reward = channels, value = num_value_hidden_channels)
template <int NSB, int CH, bool FEAT_OUT>
__global__ __launch_bounds__(256) void pre_fc_kernel(FcHead h0, FcHead h1, int NS, int ldo)
{
    const FcHead hd = blockIdx.z == 0 ? h0 : h1;
    const int nin = hd.nin, nout = hd.nout;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane >> 4, m = lane & 15;
    const int oc0 = (blockIdx.y * 4 + wave) * 16;
    if (oc0 >= nout) { return; }
    const int n0 = blockIdx.x * 16 * NSB;
    typedef __attribute__((address_space(1))) const float GF;
    const int oc = oc0 + m < nout ? oc0 + m : nout - 1; // (the last tile's rows beyond the layer repeat its last row and are dropped)
    GF* wp = (GF*)(hd.wT) + size_t(k) * nout + oc;
    GF* bp = (GF*)(hd.in) + size_t(k) * NS + n0 + m;
    const size_t wstep = size_t(4) * nout, bstep = size_t(4) * NS;
    f32x4 acc[NSB];
#pragma unroll
    for (int j = 0; j < NSB; ++j) { acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const int nsteps = nin / 4, nfull = nsteps / CH, rem = nsteps - nfull * CH; // (nin % 4 == 0)
    float ac[CH], bc[NSB][CH], an[CH], bn[NSB][CH];
    auto fetch = [&](float (&av)[CH], float (&bv)[NSB][CH], int c) { // chunk c; steps beyond the layer fetch its last step again (no branch around a load)
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int st = c * CH + s < nsteps ? c * CH + s : nsteps - 1;
            av[s] = wp[size_t(st) * wstep];
#pragma unroll
            for (int j = 0; j < NSB; ++j) { bv[j][s] = bp[size_t(st) * bstep + 16 * j]; }
        }
    };
    fetch(ac, bc, 0);
    for (int c = 0; c < nfull; ++c) {
        fetch(an, bn, c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < CH; ++s) {
#pragma unroll
            for (int j = 0; j < NSB; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[s], bc[j][s], acc[j], 0, 0, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            ac[s] = an[s];
#pragma unroll
            for (int j = 0; j < NSB; ++j) { bc[j][s] = bn[j][s]; }
        }
    }
#pragma unroll
    for (int s = 0; s < CH; ++s) { // the layer's last steps (wave-uniform count)
        if (s < rem) {
#pragma unroll
            for (int j = 0; j < NSB; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[s], bc[j][s], acc[j], 0, 0, 0); }
        }
    }
    // D: lane (k, m) holds rows oc0 + 4 * k + r (outputs), column m (sample)
    const int ob = oc0 + 4 * k;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv[r] = hd.bias[ob + r < nout ? ob + r : nout - 1]; }
#pragma unroll
    for (int j = 0; j < NSB; ++j) {
        const int n = n0 + 16 * j + m;
        if constexpr (FEAT_OUT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[j][r] + bv[r];
                v = v > 0.0f ? v : 0.0f;
                if (ob + r < nout) { hd.out[size_t(ob + r) * NS + n] = v; }
            }
        } else {
            float4 o;
            o.x = acc[j][0] + bv[0]; o.y = acc[j][1] + bv[1]; o.z = acc[j][2] + bv[2]; o.w = acc[j][3] + bv[3];
            *reinterpret_cast<float4*>(hd.out + size_t(n) * ldo + ob) = o; // (ldo = the bins padded to whole oc-tiles)
        }
    }
}

// ---- 4. per leaf: softmax expectations, policy head, candidate list, key -------------------------------------------------------------------------------------
// one workgroup of three waves per leaf, no workgroup barrier: wave 0 = reward, wave 1 = value (net_atari_body.h discreteTail's operations: maximum, exponentials,
// index-ordered sum, quotients x support, index-ordered sum; invertValue), wave 2 = policy head + candidate list + key
__global__ __launch_bounds__(192) void pre_tail_kernel(const SimArgs* __restrict__ a_, const int* __restrict__ ctl_in, const float* __restrict__ lg0, const float* __restrict__ lg1,
                                                       int ldo, int epoch)
{
    CSimArgs* a = (CSimArgs*)a_;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int4 c = reinterpret_cast<const int4*>(ctl_in)[blockIdx.x];
    if (c.x == 0) { return; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t e = size_t(c.w);
    const AtariHeadParams hp = ldc(&a->ahp);
    const int sizemax = hp.value.size > hp.reward.size ? hp.value.size : hp.reward.size, A = a->A;
    // LDS: [2][sizemax] the two heads' bins | xs [C * P] | pf [PC * P] | lgp [A] | policy [A] | logit [A] | the candidate sort's scratch
    float* xs = smem + 2 * size_t((sizemax + 3) & ~3);
    { // the rescaled state (pre_tower_kernel stored it in the leaf's slab slot) for the policy head: one coalesced trip by all three waves
        const float4* src = reinterpret_cast<const float4*>(a->hidden + e * size_t(hp.C) * hp.P);
        for (int i = threadIdx.x; i < hp.C * hp.P / 4; i += 192) { reinterpret_cast<float4*>(xs)[i] = src[i]; }
    }
    __syncthreads();
    if (wave < 2) {
        const DiscreteParams& d = wave == 0 ? hp.reward : hp.value;
        const int size = d.size;
        float* lg = smem + size_t(wave) * ((sizemax + 3) & ~3);
        const float* src = (wave == 0 ? lg0 : lg1) + size_t(blockIdx.x) * ldo;
        float m = -3.4e38f;
        for (int o = lane; o < size; o += 64) { const float v = src[o]; lg[o] = v; m = v > m ? v : m; }
        for (int o = 32; o > 0; o >>= 1) { const float m2 = __shfl_xor(m, o); m = m2 > m ? m2 : m; }
        for (int o = lane; o < size; o += 64) { lg[o] = mz_expf(lg[o] - m); }
        waveSync();
        const float s = orderedSumWave(lg, size, lane);
        waveSync();
        const int start_value = -size / 2;
        for (int o = lane; o < size; o += 64) { lg[o] = (lg[o] / s) * static_cast<float>(start_value + o); }
        waveSync();
        const float ex = orderedSumWave(lg, size, lane);
        if (lane == 0) { (wave == 0 ? a->pre_reward : a->pre_value)[e] = invertValueDev(ex); }
        return;
    }
    float* pf = xs + ((hp.C * hp.P + 3) & ~3);   // [PC * P]
    float* lgp = pf + ((hp.PC * hp.P + 3) & ~3); // [A]
    float* pol = lgp + ((A + 3) & ~3);           // [A] the raw outputs stay in LDS: what goes to the entry is the sorted list
    float* lgt = pol + ((A + 3) & ~3);           // [A]
    Cand* cs = reinterpret_cast<Cand*>(lgt + ((A + 3) & ~3));
    policyHeadWave(hp, xs, pf, lgp, pol, lgt, 0, lane);
    waveSync();
    // the leaf's candidate list in the reference's order (sim.hip simMzCandGather + orderCandidates of a non-root leaf: all A actions)
    Cand* out = cs + A;
    for (int i = lane; i < A; i += 64) { cs[i] = Cand{i, pol[i], lgt[i]}; }
    waveSync();
    orderCandidates(cs, out, reinterpret_cast<int*>(out + A), A, lane, a->err);
    waveSync();
    for (int i = lane; i < A; i += 64) {
        a->pre_action[e * A + i] = out[i].action;
        a->pre_policy[e * A + i] = out[i].policy;
        a->pre_logit[e * A + i] = out[i].logit;
    }
    if (lane == 0) {
        int* key = a->pre_key + e * 4;
        key[0] = c.y; key[1] = c.z; key[2] = epoch; key[3] = 0;
        if (a->pre_stat) { atomicAdd(a->pre_stat + 1, 1u); }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------------------
template <int CDYN_PAD, int CPAD, int NL>
static int launchPreTower(const SimArgs* d_args, const int* ctl, int nleaves, float* fT, int NS, size_t n1, hipStream_t s)
{
    using G = StackGeo<6, 6, NL>;
    const size_t tile_bytes = size_t(CDYN_PAD + CPAD) * G::CS * sizeof(float);
    const size_t epi = (size_t(NL) * CPAD * G::P + 32 + size_t(NL) * 2 * n1) * sizeof(float); // lives in T0
    if (epi > size_t(CDYN_PAD) * G::CS * sizeof(float) || tile_bytes > size_t(160) * 1024) { return MZ_ERR_ARG; }
    MZ_LDS_ATTR((pre_tower_kernel<6, 6, CDYN_PAD, CPAD, NL>), tile_bytes);
    hipLaunchKernelGGL((pre_tower_kernel<6, 6, CDYN_PAD, CPAD, NL>), dim3((nleaves + NL - 1) / NL), dim3(512), tile_bytes, s, d_args, ctl, nleaves, fT, NS);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}
template <int CDYN_PAD, int CPAD>
static int launchPreTowerNL(int NL, const SimArgs* d_args, const int* ctl, int nleaves, float* fT, int NS, size_t n1, hipStream_t s)
{
    return NL == 4 ? launchPreTower<CDYN_PAD, CPAD, 4>(d_args, ctl, nleaves, fT, NS, n1, s)
         : NL == 2 ? launchPreTower<CDYN_PAD, CPAD, 2>(d_args, ctl, nleaves, fT, NS, n1, s)
                   : launchPreTower<CDYN_PAD, CPAD, 1>(d_args, ctl, nleaves, fT, NS, n1, s);
}

template <int NSB, int CH, bool FEAT_OUT>
static int launchPreFc(const FcHead& h0, const FcHead& h1, int NS, int ldo, hipStream_t s)
{
    const int nout = std::max(h0.nout, h1.nout);
    hipLaunchKernelGGL((pre_fc_kernel<NSB, CH, FEAT_OUT>), dim3(NS / (16 * NSB), ((nout + 15) / 16 + 3) / 4, 2), dim3(256), 0, s, h0, h1, NS, ldo);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

void Net::dumpRoundsProf()
{
    unsigned long long h[3][3];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trunk_clk), sizeof(h)) != hipSuccess) { return; }
    for (int k = 0; k < 3; ++k) {
        if (h[k][2] == 0) { continue; }
        const double us = double(h[k][1]) / double(h[k][2]) * 0.01, cyc = double(h[k][0]) / double(h[k][2]);
        fprintf(stderr, "[mz sim prof] trunks of %d leaves per workgroup: %.1f us and %.0f shader cycles from staging to epilogue (workgroup 0, %llu launches) -> shader clock %.2f GHz\n",
                1 << k, us, cyc, h[k][2], cyc / us * 1e-3);
    }
}

bool Net::hasPreBatch() const
{
    if (desc_.type != 2) { return false; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    AtariHeadParams hp;
    makeAtariHeadParams(&hp);
    const int n1r = hp.reward.hc * hp.P, n1v = hp.value.hc * hp.P;
    // the instances: 6x6 boards, 64 (BASELINE configs[4]) or 32 (test nets) channels + 18 action planes; both 601-bin heads of one shape
    return H == 6 && W == 6 && (C == 64 || C == 32) && desc_.num_action_feature_channels == 18 && n1r == n1v && n1r % 4 == 0 &&
           hp.reward.hidden % 4 == 0 && hp.value.hidden % 4 == 0 && hp.reward.size == hp.value.size;
}

// the batched pipeline for the R simulations s0 .. s0 + R - 1 of every game; *launched = false: no instance (the caller falls back to sim_pre_kernel_mz)
int Net::simPreEvalBatchMz(int games, int max_depth, int s0, int R, int epoch, bool* launched, int force_nl, bool want_alt)
{
    *launched = false;
    if (!hasPreBatch() || R < 1 || epoch == 0 || pre_key_.n == 0 || sim_args_host_.size() != sizeof(SimArgs)) { return MZ_OK; }
    const SimArgs& a = *reinterpret_cast<const SimArgs*>(sim_args_host_.data());
    if (!a.pre_key || !a.use_gumbel) { return MZ_OK; }
    TowerArgs t2;
    int cd = 0;
    const int C = desc_.num_hidden_channels;
    if (!makeTowerArgs(dyn_, false, true, &t2, &cd) || !((C == 64 && cd == 84) || (C == 32 && cd == 52))) { return MZ_OK; }
    const AtariHeadParams& hp = a.ahp;
    const int n1 = hp.reward.hc * hp.P, hidden = std::max(hp.reward.hidden, hp.value.hidden), size = hp.reward.size;
    // the second expected leaf of every simulation rides along where the round leaves half of the CUs idle (the rounds of two on a pool of 64 games) and in the
    // rounds the worker has seen miss (want_alt, Worker::adaptRounds: a round with one miss costs every game of the pool a whole evaluation's latency)
    const int NH = (a.alt_base && (2 * games * R <= cu_count_ || want_alt)) ? 2 : 1;
    const int nleaves = games * R * NH;
    // leaves per tower workgroup: as many as keep every CU busy
    const int NL = (force_nl == 1 || force_nl == 2 || force_nl == 4) ? force_nl : nleaves >= 4 * cu_count_ ? 4 : nleaves >= 2 * cu_count_ ? 2 : 1;
    // A round of no more leaves than CUs is ONE leaf's latency whatever runs it, and there one workgroup per leaf doing everything (sim_pre_kernel_mz: 141 us on
    // BASELINE configs[4]) beats five dependent launches (151 us: each costs 3-4 us of launch + drain, and the FC chains cannot start before the slowest trunk
    // has ended); the pipeline takes the rounds whose leaves outnumber the CUs — there it is throughput that counts (round of 16: 538 -> 396 us, of 8: 270 -> 237 us)
    if (force_nl == 0 && NL == 1) { return MZ_OK; }
    const int NSB = force_nl > 1 ? 2 : 1; // sample tiles per wave of the FC GEMMs (2: exercised by the tests that force the leaves per workgroup)
    const int NS = (nleaves + 16 * NSB - 1) / (16 * NSB) * (16 * NSB);
    const int ldo = (size + 15) / 16 * 16;
    if (!pre_ctl_.ensure(size_t(NS) * 4 * sizeof(int)) || !pre_f_.ensure(size_t(2) * n1 * NS * sizeof(float)) || !pre_h1_.ensure(size_t(2) * hidden * NS * sizeof(float)) ||
        !pre_lg_.ensure(size_t(2) * NS * ldo * sizeof(float))) {
        setError("hipMalloc of the batched round evaluation's buffers failed");
        return MZ_ERR_DEVICE;
    }
    const SimArgs* d_args = reinterpret_cast<const SimArgs*>(sim_args_.p);
    int* ctl = reinterpret_cast<int*>(pre_ctl_.p);
    float *fT = reinterpret_cast<float*>(pre_f_.p), *h1T = reinterpret_cast<float*>(pre_h1_.p), *lg = reinterpret_cast<float*>(pre_lg_.p);
    {
        const size_t lds = (((2 * size_t(max_depth) + 2 + 3) & ~size_t(3)) * sizeof(int)) + gumbelSmemBytes(a.A);
        hipLaunchKernelGGL(pre_walk_kernel, dim3(nleaves), dim3(64), lds, stream_, d_args, s0, R, NH, ctl);
        MZ_HIP(hipGetLastError());
    }
    int rc = C == 64 ? launchPreTowerNL<84, 64>(NL, d_args, ctl, nleaves, fT, NS, n1, stream_) : launchPreTowerNL<52, 32>(NL, d_args, ctl, nleaves, fT, NS, n1, stream_);
    if (rc) { setError("simPreEvalBatchMz: the tower kernel's LDS layout does not fit"); return rc; }
    const FcHead r1{hp.reward.fc1_wT, hp.reward.fc1_b, fT, h1T, n1, hp.reward.hidden}, v1{hp.value.fc1_wT, hp.value.fc1_b, fT + size_t(n1) * NS, h1T + size_t(hidden) * NS, n1, hp.value.hidden};
    const FcHead r2{hp.reward.fc2_wT, hp.reward.fc2_b, h1T, lg, hp.reward.hidden, size}, v2{hp.value.fc2_wT, hp.value.fc2_b, h1T + size_t(hidden) * NS, lg + size_t(NS) * ldo, hp.value.hidden, size};
    if (NSB == 2) {
        if ((rc = launchPreFc<2, 17, true>(r1, v1, NS, 0, stream_))) { return rc; }
        if ((rc = launchPreFc<2, 16, false>(r2, v2, NS, ldo, stream_))) { return rc; }
    } else {
        if ((rc = launchPreFc<1, 51, true>(r1, v1, NS, 0, stream_))) { return rc; }
        if ((rc = launchPreFc<1, 16, false>(r2, v2, NS, ldo, stream_))) { return rc; }
    }
    {
        const size_t lds = (2 * size_t((size + 3) & ~3) + ((hp.C * hp.P + 3) & ~3) + ((hp.PC * hp.P + 3) & ~3) + 3 * size_t((a.A + 3) & ~3)) * sizeof(float) + azCandSmemBytes(a.A);
        hipLaunchKernelGGL(pre_tail_kernel, dim3(nleaves), dim3(192), lds, stream_, d_args, ctl, lg, lg + size_t(NS) * ldo, ldo, epoch);
        MZ_HIP(hipGetLastError());
    }
    *launched = true;
    return MZ_OK;
}

} // namespace mz
