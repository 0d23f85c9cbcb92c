// sim_kernel_mz_wide — the MuZero per-game simulation kernel (sim.hip sim_kernel_mz, board games: no leaf environment, the leaf is evaluated from its parent's
// hidden state and the move) on the ONE-TILE tower of net_wide_body.h, for the shapes whose two activation tiles do not fit the LDS: 128 / 256 hidden channels,
// 7x7 / 13x13 / 19x19 boards.  The phase functions are sim_kernel_mz's (sim_mz_body.h: same bits).  PUCT or Gumbel roots; the leaves of a Gumbel round are not
// evaluated ahead here (Net::hasPreBoardWide == false: the worker leaves mz_sim_rounds_board off for these shapes).
//   LDS: [ the tower's tile — the tree phases' scratch while no tower runs ] [ reciprocal table ] [ sqrt / bias tables + path speculation: if they fit ] [ heads scratch ]
//   global: the tower's per-game block x (SimArgs::act); the hidden states in the caller's slab as ever.
// ref muzero_network.h:97-178, zero_actor.cpp:215-245, muzero_network.py:32,81-88,137-164
#include "sim_mz_body.h"
#include "net_wide_body.h"

namespace mz {

MZ_SPEC_WAYS_IS(16); // (the launch code of this unit computes LDS sizes from kSpecWords)

template <int H, int W, int CIN0Q, int CDYNQ, int C>
__global__ __launch_bounds__(512) void sim_kernel_mz_wide(const SimArgs* __restrict__ a_, int sim0, int nsims, int host_start, int lf)
{
    using G = WideGeo<H, W, C>;
    CSimArgs* a = (CSimArgs*)a_;
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int CM = CIN0Q > CDYNQ ? (CIN0Q > C ? CIN0Q : C) : (CDYNQ > C ? CDYNQ : C);
    constexpr int kTileFloats = CM * G::CS;
    constexpr size_t CP = size_t(C) * H * W;
    double* rcp_w = reinterpret_cast<double*>(tiles + kTileFloats);
    const int rcp_n = a->rcp_n;
    for (int i = tid; i < rcp_n; i += 512) { rcp_w[i] = a->pv.rcp_tab[i]; }
    LdsCDouble* rcp_lds = (LdsCDouble*)rcp_w;
    SpecMem spec{nullptr, nullptr, nullptr};
    float* hscr = reinterpret_cast<float*>(rcp_w + rcp_n);
    if (lf & 1) { // path-speculation memory of the walk (pool_body.h): LDS copies of the sqrt / bias tables + the remembered paths
        const int tab_n = rcp_n - 2;
        double* sqrt_w = rcp_w + rcp_n;
        float* bias_w = reinterpret_cast<float*>(sqrt_w + tab_n);
        int* spec_w = reinterpret_cast<int*>(bias_w + tab_n + (tab_n & 1));
        for (int i = tid; i < tab_n; i += 512) { sqrt_w[i] = a->pv.sqrt_tab[i]; bias_w[i] = a->pv.bias_tab[i]; }
        if (tid < kSpecWays) { spec_w[tid * kSpecWay] = 0; }
        if (tid < 8) { spec_w[kSpecWays * kSpecWay + tid] = 0; }
        if (tid < kHelpSegs) { spec_w[kSpecHelp + tid * kHelpSeg] = 0; }
        spec = SpecMem{(a->no_spec & 1) ? nullptr : (LdsI32*)spec_w, (LdsCFloat*)bias_w, (LdsCDbl*)sqrt_w};
        hscr = reinterpret_cast<float*>(spec_w + kSpecWords);
    }
    const PoolView v = ldc(&a->pv);
    __syncthreads();
    unsigned long long* prof = a->prof ? a->prof + size_t(g) * 8 : nullptr;
    for (int s = 0; s < nsims; ++s) {
        const int slot = sim0 + s; // simulation index within the move = hidden-state slot of its leaf
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (prof) { t0 = wall_clock64(); }
        if (wave == 0) { simMzSelect(a, slot, s == 0 && (host_start & 1) != 0, g, lane, tiles, rcp_lds, spec, false, (host_start & 4) != 0, (a->no_spec & 2) ? 0 : s + 1); }
        else if (wave <= kHelpSegs && spec.w && !(a->no_spec & 2)) { simSelectHelper(a, g, lane, wave, s + 1, rcp_lds, spec); }
        __syncthreads();
        if (prof) { t1 = wall_clock64(); }
        float* xt = nullptr;
        if (slot == 0) { // initial inference: representation trunk on the root planes
            xt = wideTowerBody<H, W, CIN0Q, C>(reinterpret_cast<const float*>(a->root_feat), a->params, *(const TowerArgs*)&a->ta, a->act + size_t(g) * CP, a->act2 + size_t(g) * CP, g,
                                               tid, tiles, true);
        } else { // recurrent inference: dynamics trunk on (parent hidden state, move)
            const int len = v.path_len[g];
            const int* path = v.path + size_t(g) * v.max_depth;
            const int src = v.hslot[size_t(g) * v.cap + path[len - 2]];
            const int action = v.path_action[size_t(g) * v.max_depth + len - 1];
            const float* hsrc = a->hidden + (size_t(g) * a->slots + src) * CP;
            xt = wideTowerBody<H, W, CDYNQ, C>(nullptr, a->params, *(const TowerArgs*)&a->ta_dyn, a->act + size_t(g) * CP, a->act2 + size_t(g) * CP, g, tid, tiles, true, hsrc, action);
        }
        __syncthreads();
        if (prof) { t2 = wall_clock64(); }
        {
            const HeadParams hp = ldc(&a->hp);
            float* hd = a->hidden + (size_t(g) * a->slots + slot) * CP;
            rescaleTile<H, W, G::CS>(xt, hp.C, hd, tid, hscr); // scale_hidden_state in place on the tile, the rescaled state to the slab slot of this simulation
            headsBody<(H * W + 1 > 128)>(nullptr, hp, a->policy, a->logit, a->value, nullptr, nullptr, 0, g, tid, 512, hscr + 16, xt, G::CS, G::PW);
        }
        __syncthreads();
        if (prof) { t3 = wall_clock64(); }
        __shared__ int s_cand_k;
        if (wave == 0) { simMzCandGather(a, g, lane, tiles, &s_cand_k); }
        __syncthreads();
        const int cand_k = s_cand_k;
        if (a->cand_coop) { simCandRank(a->A, cand_k, wave, lane, tiles); }
        __syncthreads();
        if (wave == 0) { simMzCandExpand(a, slot, g, lane, tiles, cand_k); }
        __syncthreads();
        if (prof && tid == 0) {
            const unsigned long long t4 = wall_clock64();
            prof[0] += t1 - t0; prof[1] += t2 - t1; prof[2] += t3 - t2; prof[3] += t4 - t3; prof[4] += 1;
        }
    }
}

template <int H, int W, int CIN0Q, int CDYNQ, int C>
static int launchSimMzWideT(const SimArgs* d_args, int games, int sim0, int nsims, int host_start, int lf, size_t lds, hipStream_t s)
{
    MZ_LDS_ATTR((sim_kernel_mz_wide<H, W, CIN0Q, CDYNQ, C>), lds);
    hipLaunchKernelGGL((sim_kernel_mz_wide<H, W, CIN0Q, CDYNQ, C>), dim3(games), dim3(512), lds, s, d_args, sim0, nsims, host_start, lf);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

// (H, W, representation input channels padded to 16, dynamics input channels (hidden + one action plane) padded to 16, hidden channels)
#define MZ_SIM_MZ_WIDE_CASES(X) \
    X(9, 9, 32, 144, 128)  \
    X(9, 9, 32, 272, 256)  /* the reference's default width (configuration.cpp:71) */ \
    X(7, 7, 32, 80, 64)    \
    X(13, 13, 32, 80, 64)  \
    X(19, 19, 32, 80, 64)  \
    X(8, 8, 16, 272, 256)  /* Othello and TicTacToe (MuZero has no leaf environment: only the board differs) with the default network */ \
    X(8, 8, 16, 144, 128)  \
    X(3, 3, 16, 272, 256)

// the LDS plan: false = no instance, or the mandatory blocks do not fit.  *lf bit 0: path speculation
bool Net::simMzWidePlan(int num_simulation, int* lf, size_t* lds, size_t* tile_bytes_out, int* c0q_out, int* cdq_out) const
{
    if (desc_.type != 1 || !use_fused_ || repr_.empty() || dyn_.empty() || desc_.num_action_feature_channels != 1) { return false; }
    TowerArgs t1, t2;
    int c0q = 0, cdq = 0;
    if (!makeWideArgs(repr_, true, &t1, &c0q) || !makeWideArgs(dyn_, false, &t2, &cdq)) { return false; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    size_t tile_bytes = 0;
#define MZ_SIM_MZ_WIDE_HAS(h, w, cin0q, cdynq, c) \
    if (H == h && W == w && c0q == cin0q && cdq == cdynq && C == c) { tile_bytes = size_t(std::max(std::max(cin0q, cdynq), c)) * WideGeo<h, w, c>::CS * sizeof(float); }
    MZ_SIM_MZ_WIDE_CASES(MZ_SIM_MZ_WIDE_HAS)
#undef MZ_SIM_MZ_WIDE_HAS
    if (tile_bytes == 0) { return false; }
    HeadParams hp;
    makeHeadParams(&hp);
    const size_t rcp_n = size_t(num_simulation) + 5, A = desc_.action_size;
    const size_t heads = (16 + size_t(hp.PC) * hp.P + hp.P + hp.VH + hp.A + 16) * sizeof(float);
    size_t scratch = std::max(azCandSmemBytes(int(A)), gumbelSmemBytes(int(A)));
    scratch = std::max(scratch, size_t(2) * (size_t(num_simulation) + 3) * sizeof(float));
    size_t need = tile_bytes + rcp_n * sizeof(double) + heads + 16;
    const size_t cap = size_t(160) * 1024;
    if (need > cap || scratch > tile_bytes) { return false; }
    int f = 0;
    const size_t spec = rcp_n * (sizeof(double) + sizeof(float)) + kSpecWords * sizeof(int) + 8;
    if (need + spec <= cap) { f |= 1; need += spec; }
    if (lf) { *lf = f; }
    if (lds) { *lds = need; }
    if (tile_bytes_out) { *tile_bytes_out = tile_bytes; }
    if (c0q_out) { *c0q_out = c0q; }
    if (cdq_out) { *cdq_out = cdq; }
    return true;
}

int Net::simLaunchMzWide(const SimArgs& a, int games, int sim0, int nsims, int host_start, int lf, size_t lds, int c0q, int cdq, bool* launched)
{
    *launched = false;
    int rc = uploadSimArgs(a);
    if (rc) { return rc; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
#define MZ_SIM_MZ_WIDE_LAUNCH(h, w, cin0q, cdynq, c) \
    if (H == h && W == w && c0q == cin0q && cdq == cdynq && C == c) { *launched = true; return launchSimMzWideT<h, w, cin0q, cdynq, c>(reinterpret_cast<const SimArgs*>(sim_args_.p), games, sim0, nsims, host_start, lf, lds, stream_); }
    MZ_SIM_MZ_WIDE_CASES(MZ_SIM_MZ_WIDE_LAUNCH)
#undef MZ_SIM_MZ_WIDE_LAUNCH
    return MZ_OK;
}

} // namespace mz
