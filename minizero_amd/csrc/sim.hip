#include "sim_az_body.h"
#include "sim_mz_body.h"

namespace mz {

MZ_SPEC_WAYS_IS(16); // (the launch code of this unit computes LDS sizes from kSpecWords)

template <int H, int W, int CIN0_PAD, int CDYN_PAD, int CPAD>
__global__ __launch_bounds__(512) void sim_kernel_mz(const SimArgs* __restrict__ a_, int sim0, int nsims, int host_start, int pre_epoch)
{
    CSimArgs* a = (CSimArgs*)a_;
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int CM = CIN0_PAD > CDYN_PAD ? (CIN0_PAD > CPAD ? CIN0_PAD : CPAD) : (CDYN_PAD > CPAD ? CDYN_PAD : CPAD);
    constexpr int kTileFloats = kTowerTiles * CM * planeStride(H, W);
    // LDS: the tower tiles | the reciprocal table | (muzero_atari) the scratch of the 601-bin heads
    double* rcp_w = reinterpret_cast<double*>(tiles + kTileFloats);
    const int rcp_n = a->rcp_n;
    for (int i = tid; i < rcp_n; i += 512) { rcp_w[i] = a->pv.rcp_tab[i]; }
    LdsCDouble* rcp_lds = (LdsCDouble*)rcp_w; // (one barrier for everything the launch keeps in LDS, below: three in a row cost every launch ~2 us, and a move of BASELINE configs[4] has thirteen)
    // path-speculation memory of the walk (pool_body.h): LDS copies of the sqrt / bias tables + the remembered paths
    const int tab_n = rcp_n - 2;
    double* sqrt_w = rcp_w + rcp_n;
    float* bias_w = reinterpret_cast<float*>(sqrt_w + tab_n);
    int* spec_w = reinterpret_cast<int*>(bias_w + tab_n + (tab_n & 1));
    for (int i = tid; i < tab_n; i += 512) { sqrt_w[i] = a->pv.sqrt_tab[i]; bias_w[i] = a->pv.bias_tab[i]; }
    if (!a->atari) { // (muzero_atari walks without path speculation: its memory holds the path block and the Gumbel state filled below, before the same barrier)
        if (tid < kSpecWays) { spec_w[tid * kSpecWay] = 0; }
        if (tid < 8) { spec_w[kSpecWays * kSpecWay + tid] = 0; }
        if (tid < kHelpSegs) { spec_w[kSpecHelp + tid * kHelpSeg] = 0; }
    }
    SpecMem spec{((a->no_spec & 1) || a->atari) ? nullptr : (LdsI32*)spec_w, (LdsCFloat*)bias_w, (LdsCDbl*)sqrt_w};
    float* head_scratch = reinterpret_cast<float*>(spec_w + kSpecWords);
    // muzero_atari (no path speculation: its memory is free): the simulation's path stays in LDS — the walk writes it, the probe, expand, backup and the
    // next Gumbel step read it — instead of going through the pool's arrays in global memory (a round trip each)
    int* lds_path = (a->atari && 2 * a->pv.max_depth + 2 <= kSpecWords) ? spec_w : nullptr;
    const PoolView v = lds_path ? simPathViewSafe(ldc(&a->pv), lds_path, g) : ldc(&a->pv);
    // ... and behind them the Gumbel root's state and what its step reads of the root's children (first_child, num_children, visit counts, logits): the step
    // that runs beside a simulation's expand + backup (below) then makes no trip to global memory (6.5 -> ~2 us: it had become the longer of the two)
    int* gum_state = nullptr;
    int* gum_kids = nullptr;
    if (lds_path && a->use_gumbel && sim0 >= 1 && 2 * a->pv.max_depth + 2 + (3 + kGumbelMaxSample) + 2 + 2 * a->A <= kSpecWords) {
        gum_state = lds_path + 2 * a->pv.max_depth + 2;
        gum_kids = gum_state + 3 + kGumbelMaxSample;
    }
    if (lds_path) { // (the game's node count for the launch: the word behind the path, simPathView)
        if (tid == 0) { lds_path[2 * a->pv.max_depth + 1] = a->pv.num_nodes[g]; }
        if (gum_state) {
            const NodeRec root = a->pv.rec[size_t(g) * a->pv.cap];
            if (tid < 3 + kGumbelMaxSample) { gum_state[tid] = a->gum.state[size_t(g) * (3 + kGumbelMaxSample) + tid]; }
            if (tid == 64) { gum_kids[0] = root.first_child; gum_kids[1] = root.num_children; }
            float* kc = reinterpret_cast<float*>(gum_kids + 2);
            for (int i = tid; i < root.num_children && i < a->A; i += 512) {
                kc[i] = a->pv.rec[size_t(g) * a->pv.cap + root.first_child + i].count;
                kc[a->A + i] = a->pv.logit[size_t(g) * a->pv.cap + root.first_child + i];
            }
        }
    }
    __syncthreads();
    // THIS game's path arrays as the kernel body reads them.  The PoolView of simPathView points `g * max_depth` elements in FRONT of the LDS block (the bodies add
    // them back); inside the non-inlined phase functions those are generic pointers and 64-bit arithmetic, but here the compiler sees that the pointer is LDS and may
    // fold the bias into a DS instruction's 16-bit offset — as soon as g * max_depth * 4 exceeds the block's own LDS offset (260 Atari-shaped games at n = 50) the
    // address leaves the LDS, the read returns nothing and the slab index built from it faults (found with 512 games on one GPU, round 5).  No bias here.
    const int* const my_path = lds_path ? lds_path : v.path + size_t(g) * v.max_depth;
    const int* const my_pact = lds_path ? lds_path + a->pv.max_depth : v.path_action + size_t(g) * v.max_depth;
    const int* const my_len = lds_path ? lds_path + 2 * a->pv.max_depth : v.path_len + g;
    unsigned long long* prof = a->prof ? a->prof + size_t(g) * 8 : nullptr; // MZ_SIM_PROF=1: [select, tower, heads, cand+expand] ticks + sims
    for (int s = 0; s < nsims; ++s) {
        const int slot = sim0 + s; // simulation index within the move = hidden-state slot of its leaf
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (prof) { t0 = wall_clock64(); }
        // host_start bit 1: the root's network outputs are given (policy / logit / value / reward arrays, hidden state in slab slot 0: the muzero_atari
        // root, whose 96x96 representation runs as stand-alone kernels) — simulation 0 is only its candidate list + expand + backup
        const bool given = slot == 0 && (host_start & 2) != 0;
        // s_gum_ahead: this simulation's Gumbel step was computed beside the previous simulation's expand + backup (below)
        __shared__ int s_gum_ahead;
        __shared__ float s_bump_cnt;
        if (wave == 0) { simMzSelect(a, slot, s == 0 && (host_start & 1) != 0, g, lane, tiles, rcp_lds, spec, s > 0 && s_gum_ahead != 0, (host_start & 4) != 0, (a->no_spec & 2) ? 0 : s + 1, lds_path, gum_state); }
        else if (wave <= kHelpSegs && spec.w && !(a->no_spec & 2)) { simSelectHelper(a, g, lane, wave, s + 1, rcp_lds, spec); }
        __syncthreads();
        if (tid == 64) { s_gum_ahead = 0; } // (wave 1 sets it again below; wave 0 looked at it before the barrier)
        // was this leaf evaluated ahead (sim_pre_kernel_mz)?  Then the tower and the heads are skipped: outputs and hidden state are in place
        __shared__ int s_pre_hit;
        bool hit = false;
        if (pre_epoch != 0 && slot >= 1 && !given) {
            if (wave == 0) {
                const int len = *my_len;
                const int* path = my_path;
                const int h = simPreProbe(a, pre_epoch, g, slot, v.hslot[size_t(g) * v.cap + path[len - 2]], my_pact[len - 1], lane);
                if (lane == 0) {
                    s_pre_hit = h;
                    s_bump_cnt = v.rec[size_t(g) * v.cap + path[1]].count; // the root child on this path, before this simulation's backup
                }
                MZ_LPROF(22);
            }
            __syncthreads();
            hit = s_pre_hit >= 0;
        }
        const int eslot = hit ? s_pre_hit : slot; // the slab slot of this leaf's hidden state
        if (prof) { t1 = wall_clock64(); }
        float* xt = nullptr;
        if (given || hit) {
        } else if (slot == 0) { // initial inference: representation trunk on the root planes (board games; muzero_atari roots never come here)
            xt = towerBody<H, W, CIN0_PAD, CPAD>(reinterpret_cast<const float*>(a->root_feat), a->params, *(const TowerArgs*)&a->ta, nullptr, g, tid, tiles);
        } else { // recurrent inference: dynamics trunk on (parent hidden state, move)
            const int len = *my_len;
            const int* path = my_path;
            const int src = v.hslot[size_t(g) * v.cap + path[len - 2]];
            const int action = my_pact[len - 1];
            const float* hsrc = a->hidden + (size_t(g) * a->slots + src) * size_t(a->hp.C) * a->hp.P;
            xt = towerBody<H, W, CDYN_PAD, CPAD, (H * W <= 36)>(nullptr, a->params, *(const TowerArgs*)&a->ta_dyn, nullptr, g, tid, tiles, hsrc, action,
                                                               a->action_planes); // 6x6 = muzero_atari: stream the tower's weights (net_body.h loadW4)
        }
        __syncthreads();
        if (prof) { t2 = wall_clock64(); }
        if (!given && !hit) { simMzHeads<H, W>(a, slot, g, tid, tiles, head_scratch, xt); }
        __syncthreads();
        if (prof) { t3 = wall_clock64(); }
        __shared__ int s_cand_k;
        int cand_k = a->A; // a leaf evaluated ahead brings its sorted candidate list along: all A actions (it is never the root)
        if (!hit) {
            if (wave == 0) { simMzCandGather(a, g, lane, tiles, &s_cand_k, lds_path); }
            __syncthreads();
            cand_k = s_cand_k;
            if (a->cand_coop) { simCandRank(a->A, cand_k, wave, lane, tiles); }
            __syncthreads();
        }
        if (wave == 0) { simMzCandExpand(a, eslot, g, lane, tiles, cand_k, given, 0, hit, lds_path); }
        else if (wave == 1 && hit && a->use_gumbel && s + 1 < nsims) {
            // The next simulation's Gumbel step beside this simulation's expand + backup (4.4 of 15.7 us per simulation on BASELINE configs[4]): that backup
            // adds one visit to the root child on this path and changes nothing else the step reads — unless the candidates have all reached their
            // budget (the halving ranks them by means): then the step says so and runs after the backup as always.  The heads' scratch is free: the
            // leaf was evaluated ahead.
            const bool done = simGumbelAhead(a, slot + 1, g, lane, head_scratch, s_bump_cnt, lds_path, gum_state, gum_kids);
            if (lane == 0) { s_gum_ahead = done ? 1 : 0; }
        }
        if (gum_kids && tid == 64 && slot >= 1 && !given) { // this simulation's visit to the root child on its path, in the launch's copy of the counts (after the step ahead read them)
            const int child = my_path[1] - gum_kids[0];
            if (child >= 0 && child < a->A) { reinterpret_cast<float*>(gum_kids + 2)[child] += 1.0f; }
        }
        __syncthreads();
        if (prof && tid == 0 && !given) {
            const unsigned long long t4 = wall_clock64();
            prof[0] += t1 - t0; prof[1] += t2 - t1; prof[2] += t3 - t2; prof[3] += t4 - t3; prof[4] += 1;
        }
    }
    if (lds_path && tid == 0) { a->pv.num_nodes[g] = lds_path[2 * a->pv.max_depth + 1]; }
    if (gum_state && tid < 3 + kGumbelMaxSample) { a->gum.state[size_t(g) * (3 + kGumbelMaxSample) + tid] = gum_state[tid]; }
}


// Root exploration noise as its own launch (the leaves of the first Gumbel round are evaluated ahead of simulation 1 and need the noisy logits;
// the simulation kernels that follow get bit 2 of host_start: already applied)
__global__ __launch_bounds__(64) void sim_root_noise_kernel(const SimArgs* __restrict__ a_)
{
    CSimArgs* a = (CSimArgs*)a_;
    if (a->root_noise) { simApplyRootNoise<2>(a, blockIdx.x, threadIdx.x); }
}

// ---- Leaves of a Gumbel ROUND evaluated side by side (muzero_atari, BASELINE configs[4]) -------------------------------------------------------------
// With a Gumbel root (ref gumbel_zero.cpp:74-119) the simulations between two halvings visit the sampled root children round-robin — fewest visits first,
// then the larger logit — so the next R simulations (R = the candidates that still have the minimum visit count) walk down R DIFFERENT root children:
// their leaves only depend on their own subtrees and on the tree-wide value bounds, which rarely move.  A pool of 64 games has one dependent chain of
// 13 layers + heads per game and simulation and leaves most of the chip idle; this kernel evaluates, for every game, the leaves those R simulations are
// EXPECTED to reach (workgroup = (game, r): the Gumbel step on a private copy of the state, the PUCT walk below candidate r with the statistics as they
// stand, dynamics trunk + heads exactly as sim_kernel_mz computes them) into the slab slot and the output entry of simulation s0 + r, tagged with
// (parent slot, action, epoch).  Nothing of the search state is touched.  The simulation kernel then runs the R simulations IN ORDER as always — true
// Gumbel step, true walk, expand, backup — and skips tower + heads whenever its leaf is the tagged one (simPreProbe); a walk that ends elsewhere
// (the bounds moved, a halving fell differently) just evaluates its leaf itself.  Records cannot change: only WHERE an evaluation ran does.
// WPE = waves per SIMD the registers are budgeted for: 2 = one workgroup per CU (256 VGPRs; the rounds that leave CUs idle: latency counts), 4 = two
// workgroups per CU (128 VGPRs; the rounds of more workgroups than CUs: one workgroup's walk, heads and layer boundaries fill behind the other's MFMAs).
// Measured on BASELINE configs[4]: the round of 16 (1024 workgroups) 601 -> 535 us, the round of 8 286 -> 261 us — two workgroups on a CU take 1.85 x
// the time of one: both stream the same 4.5 MB of weights through the CU's L1 and both towers want the same four MFMA pipes.  (The tower does not spill at
// 128 registers — what spills, 39 registers, is in the heads' chains; leaving out the layer-to-layer weight prefetch or halving the heads' prefetch depth
// changed nothing.  Starting the CU's second workgroup half a run time late, so that one's heads run beside the other's tower, changed nothing either:
// the heads' vector instructions take issue slots from the MFMAs of the same SIMD.)
template <int H, int W, int CDYN_PAD, int CPAD, int WPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, 4))) void sim_pre_kernel_mz(const SimArgs* __restrict__ a_, int s0, int R, int NH, int epoch)
{
    CSimArgs* a = (CSimArgs*)a_;
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    const int g = blockIdx.x / (R * NH), q = blockIdx.x % (R * NH), r = q % R, hyp = q / R, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int CM = CDYN_PAD > CPAD ? CDYN_PAD : CPAD;
    // (the tower's second tile only ever holds CPAD channels: what follows starts there — the tower's zero fill of 2 x CM channels runs over the control
    // words and the heads' scratch, which are dead / not yet alive then — so that two workgroups fit the CU's 160 KB)
    constexpr int kTileFloats = (CM + CPAD) * planeStride(H, W);
    static_assert(kTowerTiles == 2, "layout of the tower tiles");
    const AtariHeadParams hp = ldc(&a->ahp);
    const PoolView v = ldc(&a->pv);
    // LDS: the tower tiles | [0] ok [1] parent slot [2] action [3] start node | Gumbel state copy | the walk's path | the heads' scratch (16-byte aligned)
    int* ctl = reinterpret_cast<int*>(tiles + kTileFloats);
    int* st_l = ctl + 4;
    int* path_l = st_l + 4 + kGumbelMaxSample;
    float* head_scratch = reinterpret_cast<float*>(ctl + ((4 + 4 + kGumbelMaxSample + 2 * v.max_depth + 2 + 3) & ~3));
    const int slot = s0 + r + (hyp ? a->alt_base : 0);
    if (wave == 0) {
        const int stride = 3 + kGumbelMaxSample;
        for (int i = lane; i < stride; i += 64) { st_l[i] = a->gum.state[size_t(g) * stride + i]; }
        waveSync();
        GumbelView gl = ldc(&a->gum);
        gl.state = st_l - size_t(g) * stride; // the step sorts / halves the COPY
        (void)gumbelStepBody(v, gl, s0, g, lane, tiles);
        waveSync();
        const size_t base = size_t(g) * v.cap;
        const int fc = v.rec[base].first_child, ncand = st_l[0];
        bool ok = slot < a->slots && s0 + r < (a->alt_base ? a->alt_base : a->slots) && r < ncand && r < kGumbelMaxSample;
        if (ok) { ok = v.rec[base + fc + st_l[3 + r]].count == v.rec[base + fc + st_l[3]].count; } // still in the round of candidate 0
        ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
        if (ok) {
            if (lane == 0) { ctl[3] = fc + st_l[3 + r]; }
            waveSync();
            const PoolView pl = simPathViewSafe(v, path_l, g);
            selectBody<false>(pl, ctl + 3 - g, g, lane, v.rcp_tab);
            waveSync();
            const int len = path_l[2 * v.max_depth];
            if (hyp == 0) {
                if (lane == 0) {
                    ctl[1] = v.hslot[base + path_l[len - 2]];
                    ctl[2] = path_l[v.max_depth + len - 1];
                }
            } else {
                // Hypothesis 1.  Where the true walk leaves the expected path it does so at the LAST choice between a visited child and an unvisited one
                // (every miss traced on BASELINE configs[4] was of this kind: the backups of the round's earlier simulations move the value bounds, and the
                // closest call of a walk is "once more into the child visited last, or the next sibling"): the walk stops one level earlier, at the
                // grandparent of the expected leaf, and takes that node's first unvisited child.  Never the root: its child is the Gumbel step's choice.
                ok = len >= 4;
                if (ok) {
                    const NodeRec z = v.rec[base + path_l[len - 3]];
                    const unsigned visn = static_cast<unsigned>(z.players) >> 16; // the visited children of a node are a prefix of its sorted list
                    ok = visn != 0xFFFFu && static_cast<int>(visn) < z.num_children;
                    if (ok && lane == 0) {
                        ctl[1] = v.hslot[base + path_l[len - 3]];
                        ctl[2] = v.rec[base + z.first_child + visn].action;
                    }
                }
                ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
            }
        }
        if (lane == 0) { ctl[0] = ok ? 1 : 0; }
    }
    __syncthreads();
    if (ctl[0] == 0) { return; }
    const int src = ctl[1], action = ctl[2];
    __syncthreads(); // (the tower zeroes the tiles: the Gumbel scratch in them is done with)
    const size_t hsize = size_t(a->hp.C) * a->hp.P; // (the slab's geometry: both network types keep it in hp)
    const float* hsrc = a->hidden + (size_t(g) * a->slots + src) * hsize;
    float* xt = towerBody<H, W, CDYN_PAD, CPAD, (H * W <= 36)>(nullptr, a->params, *(const TowerArgs*)&a->ta_dyn, nullptr, g, tid, tiles, hsrc, action, a->action_planes);
    __syncthreads();
    const size_t e = size_t(g) * a->slots + slot;
    float* hd = a->hidden + e * hsize;
    if (a->atari) {
        atariHeadsBody<256>(nullptr, xt, planeStride(H, W), W + 2, hp, a->pre_policy, a->pre_logit, a->pre_value, a->pre_reward, hd, 1, 1, static_cast<int>(e), tid, head_scratch);
    } else { // MuZero board games (round 4): rescale in place + slab store, policy and value heads exactly as sim_kernel_mz's simMzHeads computes them; no reward head
        const HeadParams bhp = ldc(&a->hp);
        rescaleTile<H, W>(xt, bhp.C, hd, tid, tiles); // tile 0 (the blocks' temporary) is free: its first words hold the reduction scratch
        headsBody(nullptr, bhp, a->pre_policy, a->pre_logit, a->pre_value, nullptr, nullptr, 0, static_cast<int>(e), tid, 512, tiles, xt, planeStride(H, W), W + 2);
        if (tid == 0) { a->pre_reward[e] = 0.0f; }
    }
    __syncthreads();
    if (wave == 0) { // the leaf's candidate list in the reference's order (simMzCandGather + orderCandidates of a non-root leaf: all A actions), in place of the raw outputs
        const int A = a->A;
        Cand* cs = reinterpret_cast<Cand*>(tiles);
        Cand* out = cs + A;
        for (int i = lane; i < A; i += 64) { cs[i] = Cand{i, a->pre_policy[e * A + i], a->pre_logit[e * A + i]}; }
        waveSync();
        orderCandidates(cs, out, reinterpret_cast<int*>(out + A), A, lane, a->err);
        waveSync();
        for (int i = lane; i < A; i += 64) {
            a->pre_action[e * A + i] = out[i].action;
            a->pre_policy[e * A + i] = out[i].policy;
            a->pre_logit[e * A + i] = out[i].logit;
        }
        waveSync();
    }
    if (tid == 0) {
        int* key = a->pre_key + e * 4;
        key[0] = src; key[1] = action; key[2] = epoch; key[3] = 0;
        if (a->pre_stat) { atomicAdd(a->pre_stat + 1, 1u); }
    }
}

} // namespace mz
#include "sim_cluster.h"
namespace mz {

// ---- a round's leaves on PAIRS of workgroups (muzero_atari rounds that leave half of the CUs idle and do not need their second expected leaves) --------------
// One leaf's latency is what such a round costs, and 93 of its 140 us are the trunk on ONE CU.  Two workgroups per leaf (cooperative launch; ids b and
// b + pad8(leaves): congruent mod 8 = the same XCD, checked through XCC_ID) split every layer by output-channel tile — member m computes oc-tiles 2 m, 2 m + 1,
// six of its waves one pixel tile each — and swap their halves through the XCD's L2 after every layer (sim_cluster.h clExchange: self-validating words, the
// phase in the sign bit).  Same k-ordered chain per output: the entries are bit-identical to sim_pre_kernel_mz's.  Both members run the walk (same inputs,
// same result); member 0 runs the heads and writes the entry.  xbuf: per leaf clusterWords(C, P) words, cleared by the host before the launch.
template <int H, int W, int CDYN_PAD, int CPAD>
__global__ __launch_bounds__(512) void sim_pre_pair_kernel_mz(const SimArgs* __restrict__ a_, int s0, int R, int leaves, int lpad, int epoch, unsigned* __restrict__ xbuf, int xwords, int set)
{
    CSimArgs* a = (CSimArgs*)a_;
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    const int member = blockIdx.x / lpad, leaf = blockIdx.x % lpad;
    if (leaf >= leaves) { return; }
    const int g = leaf / R, r = leaf % R, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The exchange blocks come in two sets: this launch uses `set` (all zero: "never written") and clears the leaf's block of the OTHER set — whose last users, the
    // previous pair launch on this stream, are done — for the next launch: no memset between two rounds.  (16-byte sc1 stores: they must be in the L2 / memory
    // before the next launch's loads, which bypass the vector cache.)
    {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4* other = reinterpret_cast<u4*>(xbuf + (size_t(1 - set) * lpad + leaf) * xwords);
        const int half = xwords / 4 / 2;
        for (int i = member * half + tid; i < (member == 0 ? half : xwords / 4); i += 512) { __builtin_nontemporal_store(u4{0u, 0u, 0u, 0u}, other + i); }
    }
    constexpr int CM = CDYN_PAD > CPAD ? CDYN_PAD : CPAD;
    constexpr int kTileFloats = kTowerTiles * CM * planeStride(H, W); // (towerBodyCluster zeroes two full tiles)
    const AtariHeadParams hp = ldc(&a->ahp);
    const PoolView v = ldc(&a->pv);
    __shared__ int s_abort;
    __shared__ int s_ctl[4];
    int* st_l = reinterpret_cast<int*>(tiles + kTileFloats);
    int* path_l = st_l + 4 + kGumbelMaxSample;
    float* head_scratch = reinterpret_cast<float*>(st_l + ((4 + kGumbelMaxSample + 2 * v.max_depth + 2 + 3) & ~3));
    const int slot = s0 + r;
    if (tid == 0) { s_abort = 0; }
    if (wave == 0) { // (sim_pre_kernel_mz, hypothesis 0: the Gumbel step on a private copy, the walk below candidate r)
        const int stride = 3 + kGumbelMaxSample;
        for (int i = lane; i < stride; i += 64) { st_l[i] = a->gum.state[size_t(g) * stride + i]; }
        waveSync();
        GumbelView gl = ldc(&a->gum);
        gl.state = st_l - size_t(g) * stride;
        (void)gumbelStepBody(v, gl, s0, g, lane, tiles);
        waveSync();
        const size_t base = size_t(g) * v.cap;
        const int fc = v.rec[base].first_child, ncand = st_l[0];
        bool ok = slot < a->slots && slot < (a->alt_base ? a->alt_base : a->slots) && r < ncand && r < kGumbelMaxSample;
        if (ok) { ok = v.rec[base + fc + st_l[3 + r]].count == v.rec[base + fc + st_l[3]].count; }
        ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
        if (ok) {
            if (lane == 0) { s_ctl[3] = fc + st_l[3 + r]; }
            waveSync();
            const PoolView pl = simPathViewSafe(v, path_l, g);
            selectBody<false>(pl, s_ctl + 3 - g, g, lane, v.rcp_tab);
            waveSync();
            const int len = path_l[2 * v.max_depth];
            if (lane == 0) {
                s_ctl[1] = v.hslot[base + path_l[len - 2]];
                s_ctl[2] = path_l[v.max_depth + len - 1];
            }
        }
        if (lane == 0) { s_ctl[0] = ok ? 1 : 0; }
    }
    __syncthreads();
    if (s_ctl[0] == 0) { return; } // (both members: same state, same walk)
    const int src = s_ctl[1], action = s_ctl[2];
    ClusterCtx c;
    c.cm = xbuf + (size_t(set) * lpad + leaf) * xwords;
    // (a partner that does not show up — two pair launches of different workers squeezed onto one GPU — is not an error: the leaf stays unevaluated, its simulation
    // evaluates it itself, and the counter tells the host to stop using pairs: Net::pairTrouble)
    int* trouble = reinterpret_cast<int*>(a->pre_stat + 129);
    c.member = member; c.C = hp.C; c.P = hp.P; c.OT = a->ta_dyn.OT; c.xseq = 0; c.abort_lds = &s_abort; c.err = trouble;
    c.mine0 = member * 32 * c.P; c.mine1 = c.mine0 + 32 * c.P;
    c.NJ = 0; c.j = 0; c.oseq = 0; c.om = nullptr; c.convw = nullptr;
    if (tid == 0) { // the two members must share an XCD (one L2), else the exchanges would read stale data
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        __hip_atomic_store(c.cm + kClXcc + member, (id & 15u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        clDrain();
        __hip_atomic_fetch_add(c.cm + kClXcc + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = clWaitGE(c.cm + kClXcc + 4, 2u);
        for (int m = 0; ok && m < 2; ++m) { ok = clLoadU(c.cm + kClXcc + m) == (id & 15u) + 1u; }
        if (!ok) { s_abort = 1; atomicExch(trouble, 92); }
    }
    __syncthreads();
    if (s_abort) { return; }
    const float* hsrc = a->hidden + (size_t(g) * a->slots + src) * size_t(hp.C) * hp.P;
    float* xt = towerBodyCluster<H, W, CDYN_PAD, CPAD, 2>(a->params, *(const TowerArgs*)&a->ta_dyn, tid, tiles, hsrc, action, a->action_planes, c);
    if (!xt || member != 0) { return; }
    __syncthreads();
    const size_t e = size_t(g) * a->slots + slot;
    float* hd = a->hidden + e * size_t(hp.C) * hp.P;
    atariHeadsBody<256>(nullptr, xt, planeStride(H, W), W + 2, hp, a->pre_policy, a->pre_logit, a->pre_value, a->pre_reward, hd, 1, 1, static_cast<int>(e), tid, head_scratch);
    __syncthreads();
    if (wave == 0) { // the leaf's candidate list in the reference's order, in place of the raw outputs (sim_pre_kernel_mz)
        const int A = a->A;
        Cand* cs = reinterpret_cast<Cand*>(tiles);
        Cand* out = cs + A;
        for (int i = lane; i < A; i += 64) { cs[i] = Cand{i, a->pre_policy[e * A + i], a->pre_logit[e * A + i]}; }
        waveSync();
        orderCandidates(cs, out, reinterpret_cast<int*>(out + A), A, lane, a->err);
        waveSync();
        for (int i = lane; i < A; i += 64) {
            a->pre_action[e * A + i] = out[i].action;
            a->pre_policy[e * A + i] = out[i].policy;
            a->pre_logit[e * A + i] = out[i].logit;
        }
        waveSync();
    }
    if (tid == 0) {
        int* key = a->pre_key + e * 4;
        key[0] = src; key[1] = action; key[2] = epoch; key[3] = 0;
        if (a->pre_stat) { atomicAdd(a->pre_stat + 1, 1u); }
    }
}

template <int H, int W, int CIN0_PAD, int CDYN_PAD, int CPAD>
static int launchSimPrePairMzT(const SimArgs* d_args, int leaves, int lpad, int s0, int R, int epoch, unsigned* xbuf, int xwords, int set, size_t lds, hipStream_t s)
{
    MZ_LDS_ATTR((sim_pre_pair_kernel_mz<H, W, CDYN_PAD, CPAD>), lds);
    // An ordinary launch: at most as many workgroups as CUs, on a stream whose previous kernel has ended, are all resident at once; a cooperative launch would
    // promise it but costs ~25 us of idle GPU around the kernel (12 of them per move took 0.76 -> 0.69 M leaf-evals/s off the cluster kernel), more than the
    // pairs gain.  Should the GPU be shared and a partner stay out, the waits are bounded and the leaf is simply left to its simulation (kernel comment).
    hipLaunchKernelGGL((sim_pre_pair_kernel_mz<H, W, CDYN_PAD, CPAD>), dim3(2 * lpad), dim3(512), lds, s, d_args, s0, R, leaves, lpad, epoch, xbuf, xwords, set);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

template <int H, int W, int CIN0_PAD, int CDYN_PAD, int CPAD>
static int launchSimMzT(const SimArgs* d_args, int games, int sim0, int nsims, int host_start, size_t lds, hipStream_t s, int pre_epoch)
{
    MZ_LDS_ATTR((sim_kernel_mz<H, W, CIN0_PAD, CDYN_PAD, CPAD>), lds);
    hipLaunchKernelGGL((sim_kernel_mz<H, W, CIN0_PAD, CDYN_PAD, CPAD>), dim3(games), dim3(512), lds, s, d_args, sim0, nsims, host_start, pre_epoch);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

template <int H, int W, int CIN0_PAD, int CDYN_PAD, int CPAD>
static int launchSimPreMzT(const SimArgs* d_args, int games, int s0, int R, int NH, int epoch, size_t lds, hipStream_t s, bool dense)
{
    if constexpr (H * W <= 36) {
        if (dense) {
            MZ_LDS_ATTR((sim_pre_kernel_mz<H, W, CDYN_PAD, CPAD, 4>), lds);
            hipLaunchKernelGGL((sim_pre_kernel_mz<H, W, CDYN_PAD, CPAD, 4>), dim3(games * R * NH), dim3(512), lds, s, d_args, s0, R, NH, epoch);
            MZ_HIP(hipGetLastError());
            return MZ_OK;
        }
    }
    MZ_LDS_ATTR((sim_pre_kernel_mz<H, W, CDYN_PAD, CPAD, 2>), lds);
    hipLaunchKernelGGL((sim_pre_kernel_mz<H, W, CDYN_PAD, CPAD, 2>), dim3(games * R * NH), dim3(512), lds, s, d_args, s0, R, NH, epoch);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

#define MZ_SIM_MZ_CASES(X) \
    X(9, 9, 20, 68, 64)  /* 9x9 Go MuZero, 64 channels (BASELINE configs[3]) */ \
    X(9, 9, 20, 12, 8)   /* small 9x9 test nets */ \
    X(6, 6, 64, 84, 64)  /* muzero_atari dynamics, 64 channels + 18 action planes (BASELINE configs[4]); the representation runs stand-alone */ \
    X(6, 6, 32, 52, 32)  /* small muzero_atari test nets */
#define MZ_SIM_MZ_CLUSTER_CASES(X) /* the instances that also exist as four-workgroup clusters */ \
    X(6, 6, 64, 84, 64) \
    X(6, 6, 32, 52, 32)
#define MZ_SIM_MZ_BOARD_PRE_CASES(X) /* MuZero board games whose Gumbel rounds are evaluated ahead (sim_pre_kernel_mz, one workgroup per leaf) */ \
    X(9, 9, 20, 68, 64) \
    X(9, 9, 20, 12, 8)

template <int H, int W, int CIN0_PAD, int CPAD, int CPL, bool BF = false>
static int launchSimT(const SimArgs* d_args, int games, const uint8_t* d_rot, int sim0, int nsims, int host_start, size_t lds, hipStream_t s)
{
    MZ_LDS_ATTR((sim_kernel<H, W, CIN0_PAD, CPAD, CPL, BF>), lds);
    hipLaunchKernelGGL((sim_kernel<H, W, CIN0_PAD, CPAD, CPL, BF>), dim3(games), dim3(512), lds, s, d_args, d_rot, sim0, nsims, host_start);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

#define MZ_SIM_CASES(X) \
    X(9, 9, 20, 64, 2)  /* 9x9 Go, 64 channels (BASELINE configs[1]) */ \
    X(9, 9, 20, 8, 2)   /* small 9x9 test nets */ \
    X(8, 8, 4, 64, 0)   /* 8x8 Othello, 64 channels (BASELINE configs[2]); CPL 0 = the Othello rules */ \
    X(8, 8, 4, 8, 0)    /* small 8x8 Othello test nets */ \
    X(3, 3, 4, 16, -1)  /* TicTacToe, 16 channels (BASELINE configs[0]); CPL -1 = the TicTacToe rules */

void Net::dumpSimProf()
{
#ifdef MZ_SIM_BPROF
    {
        unsigned long long h[20];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bp), sizeof(h)) == hipSuccess && h[16]) {
            for (int r = 0; r < 2; ++r) {
                fprintf(stderr, "[mz sim bprof] leaf part beside the heads, role %d (game 0, avg us over %llu calls): first piece %.2f, wait %.2f, legal mask %.2f, wait %.2f\n", r, h[16 + r],
                        double(h[r * 8 + 1]) / double(h[16 + r]) * 0.01, double(h[r * 8 + 2]) / double(h[16 + r]) * 0.01, double(h[r * 8 + 3]) / double(h[16 + r]) * 0.01, double(h[r * 8 + 4]) / double(h[16 + r]) * 0.01);
            }
        }
    }
#endif
#ifdef MZ_SIM_HPROF
    {
        unsigned long long h[16];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hp), sizeof(h)) == hipSuccess && h[15]) {
            fprintf(stderr, "[mz sim hprof] us per section of the heads (game 0, avg over %llu calls; board games: [1] tail wait, [5] setup, [6] conv1x1, [7] FCs, [8] FC2 / softmax):", h[15]);
            for (int i = 1; i < 15; ++i) { if (true) { fprintf(stderr, " [%d] %.2f", i, double(h[i]) / double(h[15]) * 0.01); } }
            fprintf(stderr, "\n");
        }
    }
#endif
#ifdef MZ_SIM_LPROF
    {
        unsigned long long h[32];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lp), sizeof(h)) == hipSuccess && h[31]) {
            fprintf(stderr, "[mz sim lprof] rank part %.2f us, introsort entered (x0.01 us as count proxy) %.4f, introsort+copy %.2f\n", double(h[13]) / double(h[31]) * 0.01, double(h[14]) / double(h[31]), double(h[8]) / double(h[31]) * 0.01);
            const char* nm[13] = {"", "leaf.load_parent", "leaf.apply_move+store", "leaf.hashes+liberties", "leaf.legal_mask", "leaf.planes", "(tower+heads)", "cand.gather", "cand.sort",
                                  "cand.store", "expand", "backup", "expand+backup"};
            fprintf(stderr, "[mz sim lprof] us per section of the tree phases (game 0, avg over %llu simulations):", h[31]);
            for (int i = 1; i < 13; ++i) { if (h[i]) { fprintf(stderr, " %s %.2f", nm[i], double(h[i]) / double(h[31]) * 0.01); } }
            for (int i = 15; i < 31; ++i) { if (h[i]) { fprintf(stderr, " [%d] %.2f", i, double(h[i]) / double(h[31]) * 0.01); } }
            fprintf(stderr, "\n");
        }
    }
#endif
#ifdef MZ_SIM_TPROF
    {
        unsigned long long h[64];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tp), sizeof(h)) == hipSuccess && h[63]) {
            fprintf(stderr, "[mz sim tprof] cycles per layer (game 0, avg over %llu towers):", h[63]);
            for (int i = 0; i < 14; ++i) { fprintf(stderr, " %.0f", double(h[i]) / double(h[63])); }
            fprintf(stderr, "\n");
        }
    }
#endif
    if (sim_prof_.n == 0) { return; }
    std::vector<unsigned long long> h(sim_prof_.n);
    if (hipMemcpy(h.data(), sim_prof_.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { return; }
    { double ps = 0, ac = 0, kk = 0; for (size_t g = 0; g < h.size() / 8; ++g) { ps += double(h[g * 8 + 7] >> 40); ac += double((h[g * 8 + 7] >> 20) & 0xFFFFF); kk += double(h[g * 8 + 7] & 0xFFFFF); } fprintf(stderr, "[mz sim prof] path speculation: %.0f passes, %.0f levels taken, %.0f walks that found a remembered path\n", ps, ac, kk); }
    const char* names[4] = {"select+leaf", "tower", "heads", "cand+expand"};
    const size_t G = h.size() / 8;
    double tot_all = 0, tot_max = 0;
    for (int k = 0; k < 4; ++k) {
        double sum = 0, mx = 0, sims = 0;
        for (size_t g = 0; g < G; ++g) { sum += double(h[g * 8 + k]); mx = std::max(mx, double(h[g * 8 + k])); sims += double(h[g * 8 + 4]); }
        fprintf(stderr, "[mz sim prof] %-12s avg %8.2f us per simulation (slowest game %8.2f us)\n", names[k], sum / std::max(1.0, sims) * 0.01,
                mx / std::max(1.0, sims / G) * 0.01);
        tot_all += sum / std::max(1.0, sims) * 0.01;
    }
    {
        double sel = 0, lev = 0, sims = 0, maxlev = 0;
        double helped = 0;
        const unsigned long long low = (1ull << 40) - 1;
        for (size_t g = 0; g < G; ++g) { sel += double(h[g * 8 + 5]); lev += double(h[g * 8 + 6] & low); helped += double(h[g * 8 + 6] >> 40); sims += double(h[g * 8 + 4]); maxlev = std::max(maxlev, double(h[g * 8 + 6] & low) / std::max(1.0, double(h[g * 8 + 4]))); }
        fprintf(stderr, "[mz sim prof] levels taken over from the helper waves: %.0f (%.1f %% of all levels)\n", helped, 100.0 * helped / std::max(1.0, lev));
        fprintf(stderr, "[mz sim prof] select alone avg %8.2f us, path length avg %.2f (deepest game avg %.2f) -> %.2f us per level\n", sel / std::max(1.0, sims) * 0.01,
                lev / std::max(1.0, sims), maxlev, sel / std::max(1.0, lev) * 0.01);
    }
    for (size_t g = 0; g < G; ++g) { tot_max = std::max(tot_max, double(h[g * 8] + h[g * 8 + 1] + h[g * 8 + 2] + h[g * 8 + 3]) / std::max(1.0, double(h[g * 8 + 4])) * 0.01); }
    fprintf(stderr, "[mz sim prof] total        avg %8.2f us per simulation (slowest game %8.2f us)\n", tot_all, tot_max);
}

// upper bound of the dynamic LDS a simulation kernel needs for searches of n simulations (the launch computes the exact figure)
static size_t simLdsBound(size_t tile_bytes, int n, int A, size_t head_floats)
{
    const size_t rcp_n = size_t(n) + 5, max_depth = size_t(n) + 3;
    return tile_bytes + rcp_n * (2 * sizeof(double) + sizeof(float)) + kSpecWords * sizeof(int) + (size_t(5) * (A + 1) + 26 + 18 * 12 + 2 * max_depth + 2 + head_floats) * sizeof(float) +
           size_t(kGoSeenCap) * sizeof(uint64_t);
}

// the argument block is constant between weight reloads / re-allocations: uploaded only when it changed
int Net::uploadSimArgs(const SimArgs& a)
{
    static_assert(sizeof(SimArgs) % 4 == 0, "SimArgs is copied as words");
    if (sim_args_host_.size() != sizeof(SimArgs) || memcmp(sim_args_host_.data(), &a, sizeof(SimArgs)) != 0) {
        if (!sim_args_.ensure(sizeof(SimArgs))) { setError("hipMalloc of the simulation arguments failed"); return MZ_ERR_DEVICE; }
        MZ_HIP(hipStreamSynchronize(stream_));
        MZ_HIP(hipMemcpy(sim_args_.p, &a, sizeof(SimArgs), hipMemcpyHostToDevice));
        sim_args_host_.assign(reinterpret_cast<const char*>(&a), reinterpret_cast<const char*>(&a) + sizeof(SimArgs));
    }
    return MZ_OK;
}

bool Net::hasSimKernelWide(int board_n, int env_kind, int num_simulation) const
{
    // (precision_: simLaunch takes the wide path for the f32 tower only — one predicate for "is there a kernel" and "will it be launched", so that a shape
    // that ever has both a bf16x3 tower and a wide instance falls back to the lock-step mode at init instead of failing at its first launch)
    if (desc_.type != 0 || !use_fused_ || repr_.empty() || precision_ != 0) { return false; }
    HeadParams hp;
    makeHeadParams(&hp);
    GoDevView gv{};
    gv.n = board_n; gv.P = board_n * board_n; gv.W = (gv.P + 63) / 64; gv.Ppad = 64 * gv.W; gv.A = desc_.action_size;
    const int max_depth = num_simulation + 3;
    size_t scratch = std::max(std::max(goLeafSmemBytes(gv, max_depth), azCandSmemBytes(gv.A)), gumbelSmemBytes(gv.A));
    scratch = std::max(scratch, size_t(2) * (size_t(num_simulation) + 8) * sizeof(float)); // >= the launch's 2 * bound_cap (pool.hip: bound_cap <= n + 3): never says yes to a plan the launch refuses
    return simWidePlan(board_n, env_kind, num_simulation, hp, desc_.num_input_channels, (gv.P + 31) / 32, goLeafSmemBytes(gv, max_depth), scratch, nullptr, nullptr, nullptr);
}

bool Net::hasSimKernel(int board_n, int env_kind, int num_simulation) const
{
    if (desc_.type != 0 || !use_fused_) { return false; }
    if (hasSimKernelWide(board_n, env_kind, num_simulation)) { return true; } // the one-tile tower (sim_wide.inc): wide / large-board shapes
    TowerArgs ta;
    int c0 = 0;
    if (!makeTowerArgs(repr_, true, true, &ta, &c0)) { return false; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    {
        const size_t tile_bytes = size_t(kTowerTiles) * std::max(c0, C) * planeStride(H, W) * sizeof(float);
        if (simLdsBound(tile_bytes, num_simulation, desc_.action_size, 0) > size_t(160) * 1024) { return false; }
        const size_t Wq = (size_t(board_n) * board_n + 63) / 64; // the Go leaf keeps the hashes of the path (max_depth words) in the tiles' scratch
        if (sizeof(uint64_t) * (64 * Wq + size_t(num_simulation) + 3 + 4 + 18 * Wq) + 64 * Wq * 7 > tile_bytes) { return false; }
    }
#define MZ_SIM_HAS(h, w, cin0, cpad, cpl) \
    if (H == h && W == w && c0 == cin0 && C == cpad && board_n == h && (env_kind == 2 ? -1 : env_kind == 1 ? 0 : (h * w + 63) / 64) == cpl) { return true; }
    MZ_SIM_CASES(MZ_SIM_HAS)
#undef MZ_SIM_HAS
    return false;
}

int Net::simLaunch(Pool& pool, const GoDevView& gv, float* d_policy, float* d_logit, float* d_value, const uint8_t* d_rot, int sim0, int nsims,
                   bool* launched, const float* d_root_noise, float noise_eps, int noise_kind, const GumbelView* gum, int* d_start, bool host_start)
{
    *launched = false;
    if (desc_.type != 0) { return MZ_OK; }
    SimArgs a;
    memset(&a, 0, sizeof(a)); // compared bytewise below: no indeterminate padding
    int c0 = 0;
    const bool wide = hasSimKernelWide(gv.n, gv.kind, pool.v_.max_depth - 3); // (false for the bf16x3 tower)
    if (wide) { if (!makeWideArgs(repr_, true, &a.ta, &c0)) { return MZ_OK; } }
    else if (!makeTowerArgs(repr_, true, true, &a.ta, &c0)) { return MZ_OK; }
    int rc = ensureBatch(gv.games);
    if (rc) { return rc; }
    makeHeadParams(&a.hp);
    a.pv = pool.v_;
    a.gv = gv;
    a.params = params_.p;
    a.act = act_[0].p;
    a.act2 = act_[1].p;
    a.policy = d_policy; a.logit = d_logit; a.value = d_value;
    a.cand_count = pool.d_cand_count_.p; a.cand_action = pool.d_cand_action_.p; a.cand_player = pool.d_cand_player_.p;
    a.cand_policy = pool.d_cand_policy_.p; a.cand_logit = pool.d_cand_logit_.p; a.value_io = pool.d_value_.p; a.reward_io = pool.d_reward_.p;
    a.err = pool.errFlag();
    a.rcp_n = pool.rcpEntries();
    a.root_noise = d_root_noise;
    a.noise_eps = noise_eps;
    a.noise_kind = noise_kind;
    a.use_gumbel = gum ? 1 : 0;
    if (gum) { a.gum = *gum; }
    a.start = d_start;
    a.no_spec = getenv("MZ_NO_SPEC") ? atoi(getenv("MZ_NO_SPEC")) : 0;
    if (getenv("MZ_SIM_PROF")) {
        if (sim_prof_.n == 0) {
            if (!sim_prof_.alloc(size_t(gv.games) * 8)) { setError("hipMalloc of the profile buffer failed"); return MZ_ERR_DEVICE; }
            MZ_HIP(hipMemset(sim_prof_.p, 0, sim_prof_.n * sizeof(unsigned long long)));
        }
        a.prof = sim_prof_.p;
    }
    if (wide) { // sim_kernel_wide: its own LDS plan (sim_wide_a.hip)
        size_t scratch = std::max(std::max(goLeafSmemBytes(gv, pool.v_.max_depth), azCandSmemBytes(gv.A)), gumbelSmemBytes(gv.A));
        scratch = std::max(scratch, size_t(2) * pool.v_.bound_cap * sizeof(float));
        int lf = 0;
        size_t lds = 0, tile_bytes = 0;
        if (!simWidePlan(gv.n, gv.kind, pool.v_.max_depth - 3, a.hp, gv.channels, gv.W32, goLeafSmemBytes(gv, pool.v_.max_depth), scratch, &lf, &lds, &tile_bytes)) { return MZ_OK; }
        a.cand_coop = (gv.A > kCandCoopMax && gv.A <= kCandCoopMaxW && candCoopSmemBytesW(gv.A, 8) <= tile_bytes) ? 2 : candCoopSmemBytes(gv.A, 8) <= tile_bytes ? 1 : 0;
        return simLaunchWide(a, gv, pool.v_.max_depth, d_rot, sim0, nsims, host_start, lf, lds, launched);
    }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    const int cmax = c0 > C ? c0 : C;
    const bool bf = precision_ == 1;
    if (bf) {
        if (!makeTowerArgsBf16(&a.tb)) { setError("simLaunch: bf16x3 tower not available for this network"); return MZ_OK; }
        a.wfrag = wfrag_.p;
    }
    // LDS: the tower tiles, and above them the reciprocal table; the heads and wave 0's tree phases take their scratch from the tiles
    const size_t tile_bytes = bf ? size_t(4) * ((((H + 2) * (W + 2) + 16) / 16) * 16) * 128 + size_t(64) * planeStride(H, W) * 4
                                 : size_t(kTowerTiles) * cmax * planeStride(H, W) * sizeof(float);
    const size_t heads = (size_t(a.hp.PC) * a.hp.P + a.hp.P + a.hp.VH + a.hp.A + 16) * sizeof(float); // in tile 0 (the activations stay in tile 1)
    size_t scratch = std::max(std::max(goLeafSmemBytes(gv, pool.v_.max_depth), azCandSmemBytes(gv.A)), gumbelSmemBytes(gv.A));
    scratch = std::max(scratch, size_t(2) * pool.v_.bound_cap * sizeof(float));
    a.cand_coop = candCoopSmemBytes(gv.A, 8) <= tile_bytes ? 1 : 0;
    if (scratch > tile_bytes || heads > tile_bytes / kTowerTiles) { // not launched
        setError("simLaunch: the tree phases need %zu B / the heads %zu B of scratch, the tower tiles have %zu B", scratch, heads, tile_bytes);
        return MZ_OK;
    }
    // (+ the path-speculation memory of the one-game-per-CU kernels: same condition as simWavesPerEu() == 2)
    const bool two_per_cu = !bf && H * W <= 64 && tile_bytes <= size_t(76) * 1024;
    const size_t lds = tile_bytes + size_t(a.rcp_n) * sizeof(double) +
                       (two_per_cu ? 0 : size_t(a.rcp_n) * (sizeof(double) + sizeof(float)) + kSpecWords * sizeof(int)) + (simXchgWords(gv.A, gv.channels, gv.W32) + 2 + 2 * size_t(pool.v_.max_depth) + 2) * sizeof(float) +
                       ((gv.kind == 0 && !two_per_cu) ? size_t(kGoSeenCap) * sizeof(uint64_t) + ((goLeafSmemBytes(gv, pool.v_.max_depth) + 7) & ~size_t(7)) : 0);
    // the argument block is constant between weight reloads / re-allocations: upload it only when it changed
    static_assert(sizeof(SimArgs) % 4 == 0, "SimArgs is copied as words");
    if (sim_args_host_.size() != sizeof(SimArgs) || memcmp(sim_args_host_.data(), &a, sizeof(SimArgs)) != 0) {
        if (!sim_args_.ensure(sizeof(SimArgs))) { setError("hipMalloc of the simulation arguments failed"); return MZ_ERR_DEVICE; }
        MZ_HIP(hipStreamSynchronize(stream_));
        MZ_HIP(hipMemcpy(sim_args_.p, &a, sizeof(SimArgs), hipMemcpyHostToDevice));
        sim_args_host_.assign(reinterpret_cast<const char*>(&a), reinterpret_cast<const char*>(&a) + sizeof(SimArgs));
    }
    if (lds > size_t(160) * 1024) { setError("simLaunch: %zu bytes of LDS needed", lds); return MZ_OK; }
    if (bf) { // the two BASELINE shapes the bf16x3 tower is built for
        if (H == 9 && W == 9 && c0 == 20 && C == 64 && gv.n == 9 && gv.kind == 0 && gv.W == 2) { *launched = true; return launchSimT<9, 9, 20, 64, 2, true>(reinterpret_cast<const SimArgs*>(sim_args_.p), gv.games, d_rot, sim0, nsims, host_start ? 1 : 0, lds, stream_); }
        if (H == 8 && W == 8 && c0 == 4 && C == 64 && gv.n == 8 && gv.kind == 1) { *launched = true; return launchSimT<8, 8, 4, 64, 0, true>(reinterpret_cast<const SimArgs*>(sim_args_.p), gv.games, d_rot, sim0, nsims, host_start ? 1 : 0, lds, stream_); }
        return MZ_OK;
    }
#define MZ_SIM_LAUNCH(h, w, cin0, cpad, cpl) \
    if (H == h && W == w && c0 == cin0 && C == cpad && gv.n == h && (gv.kind == 2 ? -1 : gv.kind == 1 ? 0 : gv.W) == cpl) { *launched = true; return launchSimT<h, w, cin0, cpad, cpl>(reinterpret_cast<const SimArgs*>(sim_args_.p), gv.games, d_rot, sim0, nsims, host_start ? 1 : 0, lds, stream_); }
    MZ_SIM_CASES(MZ_SIM_LAUNCH)
#undef MZ_SIM_LAUNCH
    return MZ_OK;
}

// MuZero board games whose Gumbel rounds can be evaluated ahead (sim_pre_kernel_mz instances; the one-tile kernels of sim_wide_mz.hip have none)
bool Net::hasPreBoard() const
{
    if (desc_.type != 1 || !use_fused_) { return false; }
    TowerArgs t1, t2;
    int c0 = 0, cd = 0;
    if (!makeTowerArgs(repr_, true, true, &t1, &c0) || !makeTowerArgs(dyn_, false, true, &t2, &cd)) { return false; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
#define MZ_SIM_PRE_HAS_B(h, w, cin0, cdyn, cpad) \
    if (H == h && W == w && c0 == cin0 && cd == cdyn && C == cpad) { return true; }
    MZ_SIM_MZ_BOARD_PRE_CASES(MZ_SIM_PRE_HAS_B)
#undef MZ_SIM_PRE_HAS_B
    return false;
}

bool Net::hasSimKernelMz(int num_simulation) const
{
    if ((desc_.type != 1 && desc_.type != 2) || !use_fused_) { return false; }
    if (simMzWidePlan(num_simulation, nullptr, nullptr, nullptr, nullptr, nullptr)) { return true; } // board games on the one-tile tower (sim_wide_mz.hip)
    TowerArgs t1, t2;
    int c0 = 0, cd = 0;
    if (desc_.type == 2) { c0 = desc_.num_hidden_channels; } // the root's representation never runs in the kernel: the CIN0_PAD = C instance
    else if (!makeTowerArgs(repr_, true, true, &t1, &c0)) { return false; }
    if (!makeTowerArgs(dyn_, false, true, &t2, &cd)) { return false; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    {
        size_t head_floats = 0;
        if (desc_.type == 2) { AtariHeadParams hp; makeAtariHeadParams(&hp); head_floats = atariHeadsSmemFloats(hp); }
        else { head_floats = (gumbelSmemBytes(desc_.action_size) + 3) / 4; } // the block simLaunchMz adds with Gumbel rounds (mode.rounds): a shape is only offered if it fits WITH it
        const size_t tile_bytes = size_t(kTowerTiles) * std::max(std::max(c0, cd), C) * planeStride(H, W) * sizeof(float);
        if (simLdsBound(tile_bytes, num_simulation, desc_.action_size, head_floats) > size_t(160) * 1024) { return false; }
    }
#define MZ_SIM_MZ_HAS(h, w, cin0, cdyn, cpad) \
    if (H == h && W == w && c0 == cin0 && cd == cdyn && C == cpad) { return true; }
    MZ_SIM_MZ_CASES(MZ_SIM_MZ_HAS)
#undef MZ_SIM_MZ_HAS
    return false;
}

int Net::simLaunchMz(Pool& pool, float* d_hidden, int slots, const unsigned* d_root_feat, const unsigned long long* d_root_legal, const int* d_root_turn,
                     int num_players, float* d_policy, float* d_logit, float* d_value, float* d_reward, int sim0, int nsims, bool* launched,
                     const float* d_root_noise, float noise_eps, int noise_kind, const GumbelView* gum, int* d_start, bool host_start, bool root_given,
                     int pre_epoch, bool noise_applied, const SimMzMode& mode)
{
    *launched = false;
    const bool atari = desc_.type == 2;
    if (desc_.type != 1 && !atari) { return MZ_OK; }
    if (root_given && (sim0 != 0 || nsims != 1)) { setError("simLaunchMz: a given root is simulation 0 alone"); return MZ_ERR_ARG; }
    if (atari && sim0 < 1 && !root_given) { setError("simLaunchMz: the muzero_atari root is evaluated by the stand-alone kernels"); return MZ_ERR_ARG; }
    SimArgs a;
    memset(&a, 0, sizeof(a));
    int c0 = 0, cd = 0;
    int wide_lf = 0;
    size_t wide_lds = 0, wide_tile = 0;
    const bool wide = !atari && precision_ == 0 && simMzWidePlan(pool.v_.max_depth - 3, &wide_lf, &wide_lds, &wide_tile, &c0, &cd);
    if (wide) {
        if (!makeWideArgs(repr_, true, &a.ta, &c0) || !makeWideArgs(dyn_, false, &a.ta_dyn, &cd)) { return MZ_OK; }
    } else {
        if (atari) { c0 = desc_.num_hidden_channels; }
        else if (!makeTowerArgs(repr_, true, true, &a.ta, &c0)) { return MZ_OK; }
        if (!makeTowerArgs(dyn_, false, true, &a.ta_dyn, &cd)) { return MZ_OK; }
    }
    int rc = ensureBatch(pool.v_.games);
    if (rc) { return rc; }
    size_t head_floats = 0;
    if (atari) {
        makeAtariHeadParams(&a.ahp);
        a.atari = 1;
        a.action_planes = desc_.num_action_feature_channels;
        a.hp.C = a.ahp.C; a.hp.P = a.ahp.P; // the slab geometry (sim_kernel_mz reads it from hp)
        head_floats = atariHeadsSmemFloats(a.ahp);
    } else {
        makeHeadParams(&a.hp);
        a.action_planes = 1;
        // (with Gumbel rounds the next simulation's Gumbel step runs beside expand + backup of a leaf that was evaluated ahead, in the block the muzero_atari heads use)
        if (gum && mode.rounds) { head_floats = (gumbelSmemBytes(desc_.action_size) + 3) / 4; }
    }
    a.pv = pool.v_;
    a.params = params_.p;
    a.policy = d_policy; a.logit = d_logit; a.value = d_value; a.reward = d_reward;
    a.cand_count = pool.d_cand_count_.p; a.cand_action = pool.d_cand_action_.p; a.cand_player = pool.d_cand_player_.p;
    a.cand_policy = pool.d_cand_policy_.p; a.cand_logit = pool.d_cand_logit_.p; a.value_io = pool.d_value_.p; a.reward_io = pool.d_reward_.p;
    a.err = pool.errFlag();
    a.rcp_n = pool.rcpEntries();
    a.act = act_[0].p; a.act2 = act_[1].p;
    a.hidden = d_hidden; a.slots = slots;
    a.alt_base = (atari && gum && mode.alt_base > 0 && 2 * mode.alt_base <= slots) ? mode.alt_base : 0; a.root_feat = d_root_feat; a.root_legal = d_root_legal; a.root_turn = d_root_turn;
    a.A = desc_.action_size; a.LW = (desc_.action_size + 63) / 64; a.num_players = num_players;
    a.root_noise = d_root_noise;
    a.noise_eps = noise_eps;
    a.noise_kind = noise_kind;
    a.use_gumbel = gum ? 1 : 0;
    if (gum) { a.gum = *gum; }
    a.start = d_start;
    if (gum && (atari || mode.rounds)) { // leaves evaluated ahead of their simulations (sim_pre_kernel_mz): one entry per (game, slot)
        const size_t ne = size_t(pool.v_.games) * slots, A = size_t(desc_.action_size);
        if (pre_key_.n != ne * 4) {
            if (!pre_key_.alloc(ne * 4) || !pre_out_.alloc(ne * (3 * A + 2)) || !pre_stat_.alloc(512)) { setError("hipMalloc of the pre-evaluation entries failed"); return MZ_ERR_DEVICE; }
            MZ_HIP(hipMemset(pre_key_.p, 0, pre_key_.n * sizeof(int)));
            MZ_HIP(hipMemset(pre_stat_.p, 0, 512 * sizeof(unsigned)));
        }
        a.pre_key = pre_key_.p;
        a.pre_policy = pre_out_.p; a.pre_logit = pre_out_.p + ne * A;
        a.pre_value = a.pre_logit + ne * A; a.pre_reward = a.pre_value + ne;
        a.pre_action = reinterpret_cast<int*>(a.pre_reward + ne);
        a.pre_stat = pre_stat_.p;
    }
    a.no_spec = getenv("MZ_NO_SPEC") ? atoi(getenv("MZ_NO_SPEC")) : 0;
    if (getenv("MZ_SIM_PROF")) {
        if (sim_prof_.n == 0) {
            if (!sim_prof_.alloc(size_t(pool.v_.games) * 8)) { setError("hipMalloc of the profile buffer failed"); return MZ_ERR_DEVICE; }
            MZ_HIP(hipMemset(sim_prof_.p, 0, sim_prof_.n * sizeof(unsigned long long)));
        }
        a.prof = sim_prof_.p;
    }
    if (wide) { // sim_kernel_mz_wide (sim_wide_mz.hip): no leaves evaluated ahead (pre_epoch is ignored: every simulation evaluates its own leaf)
        a.cand_coop = candCoopSmemBytes(a.A, 8) <= wide_tile ? 1 : 0;
        return simLaunchMzWide(a, pool.v_.games, sim0, nsims, (host_start ? 1 : 0) | (noise_applied ? 4 : 0), wide_lf, wide_lds, c0, cd, launched);
    }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    const int cmax = std::max(std::max(c0, cd), C);
    const size_t tile_bytes = size_t(kTowerTiles) * cmax * planeStride(H, W) * sizeof(float);
    size_t scratch = std::max(azCandSmemBytes(a.A), gumbelSmemBytes(a.A));
    scratch = std::max(scratch, size_t(2) * pool.v_.bound_cap * sizeof(float));
    if (scratch > tile_bytes) { return MZ_OK; }
    a.cand_coop = candCoopSmemBytes(a.A, 8) <= tile_bytes ? 1 : 0;
    const size_t lds = tile_bytes + size_t(a.rcp_n) * sizeof(double) + size_t(a.rcp_n) * (sizeof(double) + sizeof(float)) + kSpecWords * sizeof(int) +
                       head_floats * sizeof(float);
    if (lds > 160 * 1024) { return MZ_OK; }
    // cluster mode (sim_cluster.h): four workgroups per game when the pool leaves three quarters of the CUs idle (muzero_atari instances only)
    const int gpad = (pool.v_.games + 7) / 8 * 8;
    bool cluster = atari && sim_cluster_ && mode.cluster && H * W <= 36 && kClMembers * gpad <= cu_count_ && coop_launch_;
    if (cluster && sim_cluster_checked_ != gpad) { // once per pool size: do the members of a cluster share an XCD on this device?
        if (clusterPlacementOk(gpad, stream_)) { sim_cluster_checked_ = gpad; }
        else { sim_cluster_ = false; cluster = false; } // one workgroup per game (same records)
    }
    size_t lds_cluster = lds;
    if (cluster) {
        lds_cluster = lds + (clusterHeadsSmemFloats(a.ahp) - head_floats) * sizeof(float);
        if (lds_cluster > 160 * 1024) { setError("simLaunchMz: the cluster mode needs %zu bytes of LDS", lds_cluster); return MZ_ERR_ARG; }
        const size_t words = clusterWords(C, H * W);
        const DiscreteParams &dv = a.ahp.value, &dr = a.ahp.reward;
        const int n1 = std::max(dv.hc, dr.hc) * a.ahp.P, hidm = std::max(dv.hidden, dr.hidden), sizem = std::max(dv.size, dr.size);
        const size_t ow = octetWords(n1, hidm, sizem);
        // (leaves evaluated ahead make the games of an octet skip tower + heads independently of each other: every game runs its heads alone then)
        const bool octet = sim_octet_ && !mode.rounds && pool.v_.games >= 64 && size_t(4) * (up4i(n1) + up4i(hidm)) * sizeof(float) <= tile_bytes &&
                           octetHeadFits(dv, a.ahp.P) && octetHeadFits(dr, a.ahp.P);
        const size_t total_words = size_t(pool.v_.games) * words + (octet ? 16 * ow : 0);
        if (!sim_cluster_mem_.ensure(total_words * sizeof(unsigned))) { setError("hipMalloc of the cluster exchange blocks failed"); return MZ_ERR_DEVICE; }
        a.cluster = reinterpret_cast<unsigned*>(sim_cluster_mem_.p);
        a.cluster_words = static_cast<int>(words);
        a.cluster_oct = octet ? a.cluster + size_t(pool.v_.games) * words : nullptr;
        a.oct_words = static_cast<int>(ow);
        if (getenv("MZ_SIM_PROF") && sim_args_host_.empty()) { fprintf(stderr, "[mz sim] cluster mode: %d games, octet heads %s\n", pool.v_.games, octet ? "on" : "off"); }
        if (!root_given) { MZ_HIP(hipMemsetAsync(sim_cluster_mem_.p, 0, total_words * sizeof(unsigned), stream_)); }
    }
    if (sim_args_host_.size() != sizeof(SimArgs) || memcmp(sim_args_host_.data(), &a, sizeof(SimArgs)) != 0) {
        if (!sim_args_.ensure(sizeof(SimArgs))) { setError("hipMalloc of the simulation arguments failed"); return MZ_ERR_DEVICE; }
        MZ_HIP(hipStreamSynchronize(stream_));
        MZ_HIP(hipMemcpy(sim_args_.p, &a, sizeof(SimArgs), hipMemcpyHostToDevice));
        sim_args_host_.assign(reinterpret_cast<const char*>(&a), reinterpret_cast<const char*>(&a) + sizeof(SimArgs));
    }
    if (cluster && !root_given) { // (a given root is one workgroup per game: the same argument block, no exchange)
#define MZ_SIM_MZ_CL_LAUNCH(h, w, cin0, cdyn, cpad) \
        if (h * w <= 36 && H == h && W == w && c0 == cin0 && cd == cdyn && C == cpad) { \
            const int rcl = launchSimMzClusterT<h, w, cdyn, cpad>(reinterpret_cast<const SimArgs*>(sim_args_.p), pool.v_.games, sim0, nsims, (host_start ? 1 : 0) | (noise_applied ? 4 : 0), lds_cluster, stream_, pre_epoch); \
            if (rcl <= 0) { *launched = rcl == MZ_OK; return rcl; } \
            sim_cluster_ = false; /* the GPU cannot hold 4 workgroups per game right now: one workgroup per game from here on (same records) */ \
        }
        MZ_SIM_MZ_CLUSTER_CASES(MZ_SIM_MZ_CL_LAUNCH)
#undef MZ_SIM_MZ_CL_LAUNCH
    }
#define MZ_SIM_MZ_LAUNCH(h, w, cin0, cdyn, cpad) \
    if (H == h && W == w && c0 == cin0 && cd == cdyn && C == cpad) { *launched = true; return launchSimMzT<h, w, cin0, cdyn, cpad>(reinterpret_cast<const SimArgs*>(sim_args_.p), pool.v_.games, sim0, nsims, (host_start ? 1 : 0) | (root_given ? 2 : 0) | (noise_applied ? 4 : 0), lds, stream_, pre_epoch); }
    MZ_SIM_MZ_CASES(MZ_SIM_MZ_LAUNCH)
#undef MZ_SIM_MZ_LAUNCH
    return MZ_OK;
}

// the root noise as a launch of its own (before the leaves of the first Gumbel round are evaluated ahead); the argument block is the one the last
// simLaunchMz uploaded (the root expansion of the move)
int Net::simRootNoiseMz(int games)
{
    if (sim_args_host_.size() != sizeof(SimArgs)) { setError("simRootNoiseMz: no simulation launch has set up the argument block"); return MZ_ERR_STATE; }
    hipLaunchKernelGGL(sim_root_noise_kernel, dim3(games), dim3(64), 0, stream_, reinterpret_cast<const SimArgs*>(sim_args_.p));
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

// sim_pre_kernel_mz for the R simulations s0 .. s0 + R - 1 of every game (muzero_atari with a Gumbel root); *launched = false: no instance / no entries
int Net::simPreEvalMz(int games, int max_depth, int s0, int R, int epoch, bool* launched, bool want_alt, bool pairs)
{
    *launched = false;
    if ((desc_.type != 2 && desc_.type != 1) || R < 1 || epoch == 0 || pre_key_.n == 0 || sim_args_host_.size() != sizeof(SimArgs)) { return MZ_OK; }
    const SimArgs& a = *reinterpret_cast<const SimArgs*>(sim_args_host_.data());
    if (!a.pre_key || !a.use_gumbel) { return MZ_OK; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    TowerArgs t1, t2;
    int cd = 0, c0 = C; // (muzero_atari: the CIN0_PAD = C instances, its representation runs stand-alone)
    if (desc_.type == 1 && !makeTowerArgs(repr_, true, true, &t1, &c0)) { return MZ_OK; }
    if (!makeTowerArgs(dyn_, false, true, &t2, &cd)) { return MZ_OK; }
    const int cmax = std::max(cd, C);
    const size_t tile_bytes = size_t(cmax + C) * planeStride(H, W) * sizeof(float);
    if (gumbelSmemBytes(a.A) > tile_bytes) { return MZ_OK; }
    const size_t ctl_words = (4 + 4 + kGumbelMaxSample + 2 * size_t(max_depth) + 2 + 3) & ~size_t(3);
    size_t lds = tile_bytes + ctl_words * sizeof(int) + (desc_.type == 2 ? atariHeadsSmemFloats(a.ahp) * sizeof(float) : 0);
    if (desc_.type == 1) { lds = std::max(lds, size_t(kTowerTiles) * cmax * planeStride(H, W) * sizeof(float)); } // (the tower's zero fill covers two tiles of cmax channels; the board heads' scratch is tile 0)
    if (lds > 160 * 1024 || size_t(kTowerTiles) * cmax * planeStride(H, W) * sizeof(float) > lds) { return MZ_OK; }
    // the second expected leaf of every simulation rides along where the round leaves half of the CUs idle (the rounds of two on a pool of 64 games) and the
    // worker wants it (Worker::adaptRounds: while the round's simulations keep needing it); without it the idle half of the chip shortens the trunks instead
    const int NH = (a.alt_base && 2 * games * R <= cu_count_ && want_alt) ? 2 : 1;
    if (NH == 1 && pairs && pair_ok_ && 2 * ((games * R + 7) / 8 * 8) <= cu_count_ && H * W <= 36) {
        const int leaves = games * R, lpad = (leaves + 7) / 8 * 8;
        if (pair_checked_ != lpad) { // once per launch shape: do workgroups b and b + lpad share an XCD on this device?
            if (clusterPlacementOk(lpad, stream_, 2)) { pair_checked_ = lpad; } else { pair_ok_ = false; }
        }
        const size_t words = (clusterWords(C, H * W) + 7) & ~size_t(7);
        const size_t pair_lds = size_t(kTowerTiles) * cmax * planeStride(H, W) * sizeof(float) + (ctl_words + 4) * sizeof(int) + atariHeadsSmemFloats(a.ahp) * sizeof(float);
        if (pair_ok_ && pair_lds <= size_t(160) * 1024) {
            // two sets of blocks; a launch finds its set cleared by the launch before it (the kernel clears the other set's blocks of ITS leaves), by the
            // allocation, or — when it has more leaves than that launch had — by a memset
            if (pair_lpad_ != lpad || pre_pair_mem_.n < size_t(2) * lpad * words * sizeof(unsigned)) {
                if (!pre_pair_mem_.ensure(size_t(2) * lpad * words * sizeof(unsigned))) { setError("hipMalloc of the pair exchange blocks failed"); return MZ_ERR_DEVICE; }
                MZ_HIP(hipMemsetAsync(pre_pair_mem_.p, 0, size_t(2) * lpad * words * sizeof(unsigned), stream_));
                pair_lpad_ = lpad; pair_set_ = 0; pair_clean_ = lpad;
            } else if (pair_clean_ < leaves) {
                MZ_HIP(hipMemsetAsync(pre_pair_mem_.p + size_t(pair_set_) * lpad * words * sizeof(unsigned), 0, size_t(lpad) * words * sizeof(unsigned), stream_));
            }
#define MZ_SIM_PAIR_LAUNCH(h, w, cin0, cdyn, cpad) \
            if (h * w <= 36 && H == h && W == w && c0 == cin0 && cd == cdyn && C == cpad) { \
                const int rcp = launchSimPrePairMzT<h, w, cin0, cdyn, cpad>(reinterpret_cast<const SimArgs*>(sim_args_.p), leaves, lpad, s0, R, epoch, reinterpret_cast<unsigned*>(pre_pair_mem_.p), static_cast<int>(words), pair_set_, pair_lds, stream_); \
                if (rcp) { return rcp; } \
                pair_set_ ^= 1; pair_clean_ = leaves; /* the set the NEXT launch uses: cleared for this launch's leaves */ \
                *launched = true; ++pre_pair_launches_; \
                return MZ_OK; \
            }
            MZ_SIM_MZ_CLUSTER_CASES(MZ_SIM_PAIR_LAUNCH)
#undef MZ_SIM_PAIR_LAUNCH
        }
    }
    // more workgroups than CUs: two per CU (the 128-VGPR build of the kernel), if two fit the LDS
    const bool dense = NH * games * R > cu_count_ && 2 * lds <= 160 * 1024 && !getenv("MZ_PRE_SPARSE");
#define MZ_SIM_PRE_LAUNCH(h, w, cin0, cdyn, cpad) \
    if (h * w <= 36 && H == h && W == w && c0 == cin0 && cd == cdyn && C == cpad) { *launched = true; return launchSimPreMzT<h, w, cin0, cdyn, cpad>(reinterpret_cast<const SimArgs*>(sim_args_.p), games, s0, R, NH, epoch, lds, stream_, dense); }
    MZ_SIM_MZ_CLUSTER_CASES(MZ_SIM_PRE_LAUNCH)
#undef MZ_SIM_PRE_LAUNCH
#define MZ_SIM_PRE_LAUNCH_B(h, w, cin0, cdyn, cpad) \
    if (desc_.type == 1 && H == h && W == w && c0 == cin0 && cd == cdyn && C == cpad) { *launched = true; return launchSimPreMzT<h, w, cin0, cdyn, cpad>(reinterpret_cast<const SimArgs*>(sim_args_.p), games, s0, R, NH, epoch, lds, stream_, false); }
    MZ_SIM_MZ_BOARD_PRE_CASES(MZ_SIM_PRE_LAUNCH_B)
#undef MZ_SIM_PRE_LAUNCH_B
    return MZ_OK;
}

// the counters of the leaves evaluated ahead (512 words: [0] found, [1] evaluated, [2 + s] misses of simulation s, [128] second expected leaves used, [256 + s] ... by
// simulation) into a pinned host buffer, queued behind the move's launches on the network's stream
int Net::simPreCountersAsync(unsigned* h_pinned)
{
    if (pre_stat_.n < 512) { return MZ_OK; }
    MZ_HIP(hipMemcpyAsync(h_pinned, pre_stat_.p, 512 * sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
    return MZ_OK;
}

int Net::simPreStats(unsigned* hits, unsigned* evals, unsigned* alt_hits)
{
    unsigned h[512] = {0};
    if (pre_stat_.n >= 512) { MZ_HIP(hipMemcpy(h, pre_stat_.p, sizeof(h), hipMemcpyDeviceToHost)); }
    *hits = h[0]; *evals = h[1]; *alt_hits = h[128];
    if (getenv("MZ_SIM_PROF") && h[1]) {
        fprintf(stderr, "[mz sim prof] leaves evaluated ahead %u, found %u (%u of them the second expected leaf); misses by simulation of the move:", h[1], h[0], h[128]);
        for (int i = 1; i < 126; ++i) { if (h[2 + i]) { fprintf(stderr, " %d:%u", i, h[2 + i]); } }
        fprintf(stderr, "; second expected leaves used by simulation:");
        for (int i = 1; i < 126; ++i) { if (h[256 + i]) { fprintf(stderr, " %d:%u", i, h[256 + i]); } }
        fprintf(stderr, "\n");
    }
    return MZ_OK;
}

} // namespace mz
