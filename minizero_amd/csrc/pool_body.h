// Device bodies of the search-pool kernels shared by the stand-alone kernels (pool.hip) and the per-game simulation kernel
// (sim.hip): PUCT selection and expand + backup, each run by ONE wave64 for game `g`.  Numerics contract in DESIGN.md §4.
#pragma once
#include "pool.h"
#include <cfloat>
#include <climits>

#ifndef MZ_LPROF
#define MZ_LPROF(k)
#endif

namespace mz {

// normalized mean of a visited child (ref mcts.cpp:40-53 with virtual_loss == 0, which ActorGroup never uses)
__device__ __forceinline__ float normalizedMean(const PoolView& v, float reward, float mean, float cnt, int player, int bsize, float lo, float hi)
{
    float value = reward + v.gamma * mean;
    if (v.value_rescale) {
        if (bsize < 2) { return 1.0f; }
        value = (value - lo) / (hi - lo);
        value = 2 * value - 1;
        value = value < -1.0f ? -1.0f : value; // fmax(-1, .) then fmin(1, .), exact
        value = value > 1.0f ? 1.0f : value;
    }
    value = (player == v.flipping_player) ? -value : value;
    return (value * cnt - 0.0f) / (cnt + 0.0f);
}

__device__ __forceinline__ bool better(float s1, float p1, int i1, float s2, float p2, int i2)
{
    // (score, policy) lexicographic, first index wins full ties (ref mcts.cpp:189-195)
    return (s1 > s2) || (s1 == s2 && (p1 > p2 || (p1 == p2 && i1 < i2)));
}

// one step of a DPP reduction with the PUCT order: lanes without a source lane (row edge / masked row) keep their own triple
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dppBest(float& rs, float& rp, int& ri)
{
    const float s2 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(rs), __float_as_int(rs), CTRL, ROW_MASK, 0xF, false));
    const float p2 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(rp), __float_as_int(rp), CTRL, ROW_MASK, 0xF, false));
    const int i2 = __builtin_amdgcn_update_dpp(ri, ri, CTRL, ROW_MASK, 0xF, false);
    if (better(s2, p2, i2, rs, rp, ri)) { rs = s2; rp = p2; ri = i2; }
}

// global-address-space views of pool memory: `global_load` with a scalar base instead of `flat_load` with a 64-bit VGPR address (a generic
// pointer may point to LDS or scratch, so the compiler has to use the slower flat path and per-lane 64-bit address arithmetic)
#define MZ_GLOBAL __attribute__((address_space(1)))
typedef MZ_GLOBAL const NodeRec GNodeRec;
__device__ __forceinline__ NodeRec loadRec(GNodeRec* p)
{
    typedef float vf4 __attribute__((ext_vector_type(4)));
    typedef int vi4 __attribute__((ext_vector_type(4)));
    const vf4 a = ((MZ_GLOBAL const vf4*)p)[0];
    const vi4 b = ((MZ_GLOBAL const vi4*)p)[1];
    NodeRec n;
    n.count = a.x; n.mean = a.y; n.policy = a.z; n.reward = a.w;
    n.first_child = b.x; n.num_children = b.y; n.action = b.z; n.players = b.w;
    return n;
}
__device__ __forceinline__ NodeRec loadRec(const NodeRec* p)
{
    const float4 a = reinterpret_cast<const float4*>(p)[0];
    const int4 b = reinterpret_cast<const int4*>(p)[1];
    NodeRec n;
    n.count = a.x; n.mean = a.y; n.policy = a.z; n.reward = a.w;
    n.first_child = b.x; n.num_children = b.y; n.action = b.z; n.players = b.w;
    return n;
}

// value of a wave-uniform lane: v_readlane_b32 (a few cycles) instead of the ds_bpermute_b32 that __shfl turns into (an LDS round
// trip, ~100 cycles: the ordered init-Q sum walks up to A visited children one by one)
__device__ __forceinline__ int laneI(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float laneF(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// ---- fast PUCT level (nodes of <= 128 children, no value rescaling: every board game) ----
// The walk is ONE wave executing a long dependent instruction stream per level (measured: ~390 cycles waiting for the children
// records, ~2900 cycles of arithmetic), so the level is written for instruction count:
//  * both IEEE divisions of the reference's formulas go through a host-built table of correctly rounded reciprocals of the
//    (integer) visit counts: f32 a / c == (float)((double)a * RN64(1/c)) (the double product is within 2^-52 of a/c, a float
//    rounding boundary is never that close to a quotient of a float by an integer <= 4096 unless the quotient is subnormal ->
//    guarded), f64 x / d == fma(fma(-d, q0, x), r, q0) with q0 = x * r (Markstein's correction step, exact for r = RN(1/d));
//    tests/test_div_tricks.py checks both against real division;
//  * the arg-max reduces order-preserving integer keys with one DPP max/min per step instead of a three-value compare-and-select.
__device__ __forceinline__ unsigned orderedKey(float f)
{
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); // unsigned order == float order (no NaNs, no negative zeros here)
}
// wave-wide unsigned max, result in every lane: six v_max_u32 with the DPP operand folded in (row_shr 1/2/4/8 scan inside the rows of
// 16, then row_bcast 15 / 31 across rows; lanes without a source read 0, the identity), written in assembly because the compiler
// emits v_mov_dpp + s_nop + v_max per step; the s_nop 1 is the VALU-write -> DPP-read hazard (2 wait states)
__device__ __forceinline__ unsigned waveMaxU32(unsigned v)
{
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

struct LevelEval { float q, score; };
// q = normalized mean of a visited child (ref mcts.cpp:40-53), u = PUCT exploration term (mcts.cpp:55-61); rc = {RN(1/count), RN(1/(1+count))}
__device__ __forceinline__ LevelEval evalChild(const PoolView& v, const NodeRec& c, int cplayer, float bias, double sqrtN, double rc0, double rc1, bool* tiny)
{
    float value = c.reward + v.gamma * c.mean;
    value = (cplayer == v.flipping_player) ? -value : value;
    const float a = value * c.count - 0.0f;
    LevelEval e;
    e.q = static_cast<float>(static_cast<double>(a) * rc0); // == a / (count + 0.0f) unless the quotient is subnormal
    *tiny = (a != 0.0f) && (__builtin_fabsf(a) < 0x1p-100f);
    const float bpol = bias * c.policy;
    const double x = static_cast<double>(bpol) * sqrtN;
    const double d = static_cast<double>(1 + c.count);
    const double q0 = x * rc1;
    const double y = __builtin_fma(__builtin_fma(-d, q0, x), rc1, q0); // == x / d
    e.score = static_cast<float>(y);
    return e;
}

// One wave64 per game.  Per level: every lane loads the 32-B records of its children (<= 2 per lane for A <= 128, a loop
// beyond), all arithmetic runs from registers, and the winning lane's record supplies the next level's (first_child,
// num_children, count) through shuffles — a single dependent memory round trip per level.
// rcp: the reciprocal table RN64(1/i) — v.rcp_tab (global) or an LDS copy of it (the simulation kernel: the table lookup is on the
// per-level critical path, right behind the children records)
// Path speculation (simulation kernel only; `spec` = LDS words, nullptr: off).  A deep principal variation is walked again by almost every
// simulation, one dependent level (~1 800 cycles of a single wave's instruction stream) after the other, and the launch lasts as long as
// its deepest game.  The walk remembers, per level of the previous path, the node, its children block and its visit count after that
// simulation's backup (nothing else can have changed it); while the new walk is on the previous path, 4 lanes per level evaluate the next
// 16 predicted levels in ONE pass of the same arithmetic (per-lane parent parameters), and a scalar loop then accepts level after level as
// long as the arg-max is the predicted node and the prediction checks out against the records just loaded.
// Only immutable facts are remembered (a node's children block never moves once it is expanded), so an entry can never be wrong, only
// useless; kSpecWays paths are kept (PUCT rotates through the better root children), the walk uses the one whose first move it repeats
// and overwrites the oldest one otherwise.  The per-level tables (log / sqrt of the visit count) are LDS copies: their index is only known
// once the records have arrived, and a second dependent trip to global memory would cost what the speculation saves.
typedef __attribute__((address_space(3))) int LdsI32;
typedef __attribute__((address_space(3))) const float LdsCFloat;
typedef __attribute__((address_space(3))) const double LdsCDbl;
constexpr int kSpecCap = 128;                                   // levels remembered per path
constexpr int kSpecWay = 4 + 3 * kSpecCap;                      // words of one remembered path: [0] length, then node / first_child / num_children per level
constexpr int kGw = 4, kLv = 64 / kGw;                          // lanes per predicted level (its first children) and levels per pass
#ifndef MZ_SPEC_WAYS
#define MZ_SPEC_WAYS 16 // (a translation unit whose kernels have little LDS to spare defines fewer — a power of two: sim_wide_c.hip, 19x19 Go)
#endif
// MZ_SPEC_WAYS is a PER-TRANSLATION-UNIT choice: kSpecWays and everything derived from it (kSpecHelp, kSpecWords, the layouts selectBody walks) have internal
// linkage (namespace-scope constexpr) and are only used by device code of the same unit — the library is built without relocatable device code, so no kernel or
// device function is shared between units — and by that unit's OWN host-side LDS sizing.  A host function or a kernel instantiation shared between units must
// never read them; each unit that sizes LDS from kSpecWords states the value it was built with (MZ_SPEC_WAYS_IS below).
constexpr int kSpecWays = MZ_SPEC_WAYS;                         // remembered paths (one per recently walked root child)
#define MZ_SPEC_WAYS_IS(n) static_assert(mz::kSpecWays == (n), "this translation unit sizes its LDS for " #n " remembered paths (pool_body.h MZ_SPEC_WAYS)")
static_assert((kSpecWays & (kSpecWays - 1)) == 0 && kSpecWays >= 2 && kSpecWays <= 64, "a power of two of at most one wave's lanes");
// Helper segments (selectSpecHelper): while wave 0 walks levels 1 .. 16 of a remembered path, waves 1 .. 3 evaluate levels 17 .. 32, 33 .. 48 and 49 .. 64 of the path the
// PREVIOUS walk took, each into its own result block; wave 0 takes a block over when it arrives at the block's entry node with all 16 levels before it accepted.
constexpr int kHelpSegs = 3;                                    // helper waves / segments of kLv levels behind the first
constexpr int kHelpHdr = 16;                                    // [0] serial of the simulation the block belongs to, [2] entry node, [3] levels accepted, [4..9] header of the last chosen node + the node
constexpr int kHelpSeg = kHelpHdr + 4 * kLv;                    // + per accepted level: chosen node, its action, first_child, num_children
constexpr int kSpecHelp = kSpecWays * kSpecWay + 8;
constexpr int kSpecWords = kSpecHelp + kHelpSegs * kHelpSeg;    // the paths + [kSpecWays * kSpecWay] = the next one to replace; [+1] passes, [+3] walks that found their path, [+5] levels taken, [+6] the way of the last walk, [+7] levels taken over from helpers; the helpers' blocks
constexpr int kSpecNode = 4, kSpecFc = 4 + kSpecCap, kSpecNc = 4 + 2 * kSpecCap;
struct SpecMem { LdsI32* w; LdsCFloat* bias; LdsCDbl* sqrt; }; // w == nullptr: no speculation

// serial > 0: helper waves run selectSpecHelper(serial) beside this walk (the per-game simulation kernels): their blocks are taken over where they fit
template <bool SPEC = false, class RcpPtr>
__device__ __forceinline__ void selectBody(const PoolView& v, const int* __restrict__ start, int g, int lane, RcpPtr rcp, SpecMem sm = SpecMem{nullptr, nullptr, nullptr}, int serial = 0)
{
    GNodeRec* recs = (GNodeRec*)(v.rec + size_t(g) * v.cap);
    // (generic pointers: the simulation kernel keeps the path of its game in LDS and points the view there)
    int* path = v.path + size_t(g) * v.max_depth;
    int* pact = v.path_action + size_t(g) * v.max_depth;
    int* hact = v.host_path_action ? v.host_path_action + size_t(g) * v.max_depth : nullptr;
    MZ_GLOBAL const float* bias_tab = (MZ_GLOBAL const float*)v.bias_tab;
    MZ_GLOBAL const double* sqrt_tab = (MZ_GLOBAL const double*)v.sqrt_tab;
    // remember level `d` of this walk for the next simulation: its count will be one higher after this simulation's backup
    LdsI32* spec = nullptr; // the remembered path this walk follows / overwrites: chosen at its first step below the root
    auto note = [&](int d, int n, const NodeRec& r) {
        if (!SPEC || !sm.w) { return; }
        if (d == 1) { // which remembered path starts with this move?  None: replace the oldest (round robin)
            const int w = lane < kSpecWays ? lane : 0;
            const int lw = sm.w[w * kSpecWay], nw = sm.w[w * kSpecWay + kSpecNode + 1];
            const unsigned long long m = __ballot(lane < kSpecWays && lw > 1 && nw == n);
            int way;
            if (m != 0) {
                way = static_cast<int>(__builtin_ctzll(m));
                if (lane == 0) { sm.w[kSpecWays * kSpecWay + 3] += 1; }
            } else {
                way = __builtin_amdgcn_readfirstlane(sm.w[kSpecWays * kSpecWay]) & (kSpecWays - 1);
                if (lane == 0) { sm.w[kSpecWays * kSpecWay] = way + 1; sm.w[way * kSpecWay] = 0; }
            }
            spec = sm.w + way * kSpecWay;
            if (lane == 0) { sm.w[kSpecWays * kSpecWay + 6] = way; } // (the helpers of the NEXT walk follow this path)
        }
        if (spec && lane == 0 && d < kSpecCap) {
            spec[kSpecNode + d] = n;
            spec[kSpecFc + d] = r.first_child;
            spec[kSpecNc + d] = r.num_children;
        }
    };
    const int bsize = v.bound_size[g];
    const float lo = v.bound_lo[g], hi = v.bound_hi[g];
    // the node header travels down the walk in SCALAR registers: every lane loads the same record, readfirstlane tells the compiler so
    // (otherwise the whole loop is compiled as divergent: exec-mask loop control, per-level v_mov round trips of the header)
    auto uniformRec = [](NodeRec r) {
        NodeRec u;
        u.count = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r.count)));
        u.mean = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r.mean)));
        u.policy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r.policy)));
        u.reward = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r.reward)));
        u.first_child = __builtin_amdgcn_readfirstlane(r.first_child);
        u.num_children = __builtin_amdgcn_readfirstlane(r.num_children);
        u.action = __builtin_amdgcn_readfirstlane(r.action);
        u.players = __builtin_amdgcn_readfirstlane(r.players);
        return u;
    };
    NodeRec cur = uniformRec(loadRec(recs)); // root
    int node = 0, depth = 1;
    if (lane == 0) { path[0] = 0; pact[0] = cur.action; if (hact) { hact[0] = cur.action; } }
    const int st = start ? __builtin_amdgcn_readfirstlane(start[g]) : 0;
    if (st > 0) { // Gumbel: path = root + PUCT path below the chosen candidate (ref gumbel_zero.cpp:83-85)
        node = st;
        cur = uniformRec(loadRec(recs + st));
        if (lane == 0) { path[1] = st; pact[1] = cur.action; if (hact) { hact[1] = cur.action; } }
        note(1, st, cur);
        depth = 2;
    }
    const int max_depth = __builtin_amdgcn_readfirstlane(v.max_depth);
    for (;;) {
        // the loop-carried header is wave-uniform by construction (readlane results); saying so again at the top of every level keeps the
        // loop control and the per-level branches scalar (without it the compiler builds the walk with exec-mask control flow)
        cur.count = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cur.count)));
        cur.first_child = __builtin_amdgcn_readfirstlane(cur.first_child);
        cur.num_children = __builtin_amdgcn_readfirstlane(cur.num_children);
        cur.players = __builtin_amdgcn_readfirstlane(cur.players);
        depth = __builtin_amdgcn_readfirstlane(depth);
        node = __builtin_amdgcn_readfirstlane(node);
        if (!(cur.num_children != 0 && depth < max_depth)) { break; }
        const int nc = cur.num_children, fc = cur.first_child, cplayer = (cur.players >> 8) & 0xFF;
        const int N = static_cast<int>(cur.count - 1);
        const float bias = bias_tab[N];
        const double sqrtN = sqrt_tab[N];
        // (a node of more than 128 children — 13x13 / 19x19 Go — takes the same path below the root whenever the children that need a look are at most 128:
        //  the visited-prefix argument does not depend on the node's width)
        if (nc <= 128 || (node != 0 && min(nc, static_cast<int>(static_cast<unsigned>(cur.players) >> 16) + 1) <= 128)) { // (with value rescaling the normalised means come from normalizedMean() itself: the reciprocal trick covers the plain case)
            // Children are stored in descending prior order and an unvisited child can only be chosen while every child before it has
            // been visited (equal init-Q, u monotone in the prior, ties go to the higher prior / lower index), so the visited children
            // of a node are a PREFIX of its children and the arg-max is among that prefix plus the first unvisited child.  The prefix
            // length is kept in the upper half of `players` (expandBackupBody).  Not at the root: the root noise re-orders its priors.
            const int ne = node == 0 ? nc : min(nc, static_cast<int>(static_cast<unsigned>(cur.players) >> 16) + 1); // children that need a look
            if (SPEC && spec && node != 0 && !v.atari_init_q && !v.value_rescale) {
                const int L0 = depth - 1;
                const int plen = __builtin_amdgcn_readfirstlane(spec[0]);
                // a helper wave has evaluated the 16 levels from this node on (selectSpecHelper): take its block over.  The block says "from entry node E, with the records
                // as they are, the arg-max choices are c1, c2, ..." — every level validated against the records it loaded — so it holds whatever path it was found on
                if (serial > 0 && L0 > kLv && (L0 - 1) % kLv == 0 && (L0 - 1) / kLv <= kHelpSegs) {
                    LdsI32* h = sm.w + kSpecHelp + ((L0 - 1) / kLv - 1) * kHelpSeg;
                    if (__builtin_amdgcn_readfirstlane(h[0]) == serial) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const int hadv = __builtin_amdgcn_readfirstlane(h[3]);
                        if (__builtin_amdgcn_readfirstlane(h[2]) == node && hadv > 0 && depth + hadv <= max_depth) {
                            if (lane < hadv) {
                                const int ch = h[kHelpHdr + lane], ca = h[kHelpHdr + kLv + lane];
                                path[depth + lane] = ch;
                                pact[depth + lane] = ca;
                                if (hact) { hact[depth + lane] = ca; }
                                if (depth + lane < kSpecCap) {
                                    spec[kSpecNode + depth + lane] = ch;
                                    spec[kSpecFc + depth + lane] = h[kHelpHdr + 2 * kLv + lane];
                                    spec[kSpecNc + depth + lane] = h[kHelpHdr + 3 * kLv + lane];
                                }
                            }
                            cur.count = __int_as_float(__builtin_amdgcn_readfirstlane(h[4]));
                            cur.first_child = __builtin_amdgcn_readfirstlane(h[5]);
                            cur.num_children = __builtin_amdgcn_readfirstlane(h[6]);
                            cur.action = __builtin_amdgcn_readfirstlane(h[7]);
                            cur.players = __builtin_amdgcn_readfirstlane(h[8]);
                            node = __builtin_amdgcn_readfirstlane(h[9]);
                            depth += hadv;
                            if (lane == 0) { sm.w[kSpecWays * kSpecWay + 7] += hadv; }
                            continue;
                        }
                    }
                }
                if (L0 + 1 < plen && L0 + 1 < kSpecCap && __builtin_amdgcn_readfirstlane(spec[kSpecNode + L0]) == node) {
                    int K = plen - L0 < kSpecCap - L0 ? plen - L0 : kSpecCap - L0;
                    K = K < kLv ? K : kLv;
                    const int k = lane / kGw, j = lane % kGw, Lk = L0 + k, base8 = lane - j;
                    const bool inrange = k < K;
                    const int nk = inrange ? (k == 0 ? node : spec[kSpecNode + Lk]) : 0;
                    const int fck = inrange ? (k == 0 ? fc : spec[kSpecFc + Lk]) : 0;
                    const int nck = inrange ? (k == 0 ? nc : spec[kSpecNc + Lk]) : 0;
                    const int pfc = (inrange && k >= 1) ? (k == 1 ? fc : spec[kSpecFc + Lk - 1]) : 0;   // the block the predicted node lives in
                    const int pnc = (inrange && k >= 1) ? (k == 1 ? nc : spec[kSpecNc + Lk - 1]) : 0;
                    const int nl = nck < kGw ? nck : kGw;
                    const bool ld = inrange && j < nl;
                    const NodeRec c = loadRec(recs + (ld ? fck + j : 0));
                    // the header of level k's node is one of the records level k - 1 loaded
                    const int idx = nk - pfc;
                    const bool idx_ok = idx >= 0 && idx < kGw && idx < pnc;
                    const int srcl = idx_ok ? base8 - kGw + idx : lane;
                    float hcount = __shfl(c.count, srcl);
                    int hfc = __shfl(c.first_child, srcl), hnc = __shfl(c.num_children, srcl), hpl = __shfl(c.players, srcl);
                    if (k == 0) { hcount = cur.count; hfc = fc; hnc = nc; hpl = cur.players; }
                    bool okk = inrange && nck > 0 && (k == 0 || (idx_ok && hfc == fck && hnc == nck && hcount >= 1.0f));
                    int Nk = static_cast<int>(hcount - 1);
                    Nk = okk ? Nk : 0;
                    const float biask = sm.bias ? sm.bias[Nk] : bias_tab[Nk];
                    const double sqrtNk = sm.sqrt ? sm.sqrt[Nk] : sqrt_tab[Nk];
                    const unsigned visn = static_cast<unsigned>(hpl) >> 16;
                    const int nek = (visn == 0xFFFFu) ? nck : (nck < static_cast<int>(visn) + 1 ? nck : static_cast<int>(visn) + 1);
                    okk = okk && nek <= kGw;
                    const int cplk = (hpl >> 8) & 0xFF;
                    const bool has = okk && j < nek;
                    const RcpPtr rp = rcp + (has ? static_cast<int>(c.count) : 0);
                    bool tiny = false;
                    const LevelEval e = evalChild(v, c, cplk, biask, sqrtNk, rp[0], rp[1], &tiny);
                    const bool vis = has && c.count != 0.0f;
                    if (__ballot(vis && tiny) == 0) { // (a subnormal quotient needs the reference's division: leave it to the plain walk)
                        int mx = 0;
#pragma unroll
                        for (int t = 1; t <= kGw; ++t) { if (__ballot(okk && nek >= t) != 0) { mx = t; } }
                        // init Q: the ordered f32 sum over the visited children of the level's node (adding +0 for the others changes nothing)
                        const float qm = vis ? e.q : 0.0f, vm1 = vis ? 1.0f : 0.0f;
                        float sum_of_win = 0.0f, sum = 0.0f;
                        if (mx <= 3) { // the usual case, straight-line: the six gathers are in flight together
                            const float q0 = __shfl(qm, base8), q1 = __shfl(qm, base8 + 1), q2 = __shfl(qm, base8 + 2);
                            const float v0 = __shfl(vm1, base8), v1 = __shfl(vm1, base8 + 1), v2 = __shfl(vm1, base8 + 2);
                            sum_of_win = ((sum_of_win + q0) + q1) + q2;
                            sum = ((sum + v0) + v1) + v2;
                        } else {
                            for (int i = 0; i < mx; ++i) {
                                const float qi = __shfl(qm, base8 + i), vi = __shfl(vm1, base8 + i);
                                sum_of_win = sum_of_win + qi;
                                sum = sum + vi;
                            }
                        }
                        const float init_q = (sum_of_win - 1) / (sum + 1);
                        const float sc = e.score + (c.count == 0.0f ? init_q : e.q);
                        float bs = 0.0f, bp = 0.0f;
                        int bi = 0;
                        if (mx <= 3) {
                            const float s0 = __shfl(sc, base8), s1 = __shfl(sc, base8 + 1), s2 = __shfl(sc, base8 + 2);
                            const float p0 = __shfl(c.policy, base8), p1 = __shfl(c.policy, base8 + 1), p2 = __shfl(c.policy, base8 + 2);
                            bs = s0; bp = p0; bi = 0;
                            if (1 < nek && better(s1, p1, 1, bs, bp, bi)) { bs = s1; bp = p1; bi = 1; }
                            if (2 < nek && better(s2, p2, 2, bs, bp, bi)) { bs = s2; bp = p2; bi = 2; }
                        } else {
                            for (int i = 0; i < mx; ++i) {
                                const float si = __shfl(sc, base8 + i), pi = __shfl(c.policy, base8 + i);
                                if (i < nek && (i == 0 || better(si, pi, i, bs, bp, bi))) { bs = si; bp = pi; bi = i; }
                            }
                        }
                        const int chl = base8 + bi; // the lane that holds the level's chosen child
                        const float ch_count = __shfl(c.count, chl);
                        const int ch_fc = __shfl(c.first_child, chl), ch_nc = __shfl(c.num_children, chl), ch_act = __shfl(c.action, chl),
                                  ch_pl = __shfl(c.players, chl);
                        // how many levels does the walk take?  Level k counts if levels 0 .. k - 1 did, its own data checked out (okk), the level
                        // before it chose the node it was evaluated for, and that node was not the end of the walk
                        const int chosen = fck + bi;
                        const int prev_chosen = __shfl(chosen, lane >= kGw ? lane - kGw : lane), prev_nc = __shfl(ch_nc, lane >= kGw ? lane - kGw : lane);
                        const bool link = okk && (k == 0 || (prev_chosen == nk && prev_nc != 0)) && depth + k < max_depth;
                        const unsigned long long lm = __ballot(j == 0 && !link); // bit 8k set: level k breaks the chain
                        const int adv_all = lm ? static_cast<int>(__builtin_ctzll(lm)) / kGw : kLv;
                        const int adv = adv_all < K ? adv_all : K;
                        if (adv > 0) {
                            if (j == 0 && k < adv) { // every accepted level writes its own path entry and remembers itself
                                path[depth + k] = chosen;
                                pact[depth + k] = ch_act;
                                if (hact) { hact[depth + k] = ch_act; }
                                if (depth + k < kSpecCap) {
                                    spec[kSpecNode + depth + k] = chosen;
                                    spec[kSpecFc + depth + k] = ch_fc;
                                    spec[kSpecNc + depth + k] = ch_nc;
                                }
                            }
                            const int l = kGw * (adv - 1);
                            cur.count = laneF(ch_count, l);
                            cur.first_child = laneI(ch_fc, l);
                            cur.num_children = laneI(ch_nc, l);
                            cur.action = laneI(ch_act, l);
                            cur.players = laneI(ch_pl, l);
                            node = laneI(chosen, l);
                            depth += adv;
                        }
                        if (lane == 0) { sm.w[kSpecWays * kSpecWay + 1] += 1; sm.w[kSpecWays * kSpecWay + 5] += adv; }
                        if (adv > 0) { continue; }
                    }
                }
            }
            if (node != 0 && ne <= 3) {
                // 96 % of the levels of a 400-simulation search: one or two visited children plus the first unvisited one.  Same
                // arithmetic, but the handful of values is compared through readlanes: no wave reductions, no loops.
                const bool has = lane < ne;
                const NodeRec c = loadRec(recs + fc + (has ? lane : ne - 1));
                const RcpPtr rp = rcp + static_cast<int>(c.count);
                bool tiny = false;
                LevelEval e = evalChild(v, c, cplayer, bias, sqrtN, rp[0], rp[1], &tiny);
                const bool vis = has && c.count != 0.0f;
                const unsigned long long vm = __ballot(vis);
                if (v.value_rescale || __ballot(vis && tiny) != 0) { e.q = normalizedMean(v, c.reward, c.mean, c.count, cplayer, bsize, lo, hi); }
                // ordered sum over the visited children (adding +0 for an unvisited one changes nothing: the sum is never -0)
                float sum_of_win = 0.0f, sum = 0.0f;
                sum_of_win += (vm & 1) ? laneF(e.q, 0) : 0.0f; sum += (vm & 1) ? 1.0f : 0.0f;
                sum_of_win += (vm & 2) ? laneF(e.q, 1) : 0.0f; sum += (vm & 2) ? 1.0f : 0.0f;
                sum_of_win += (vm & 4) ? laneF(e.q, 2) : 0.0f; sum += (vm & 4) ? 1.0f : 0.0f;
                const float init_q = v.atari_init_q ? (sum > 0 ? sum_of_win / sum : 1.0f) : (sum_of_win - 1) / (sum + 1);
                const float sc = e.score + (c.count == 0.0f ? init_q : e.q);
                float bs = laneF(sc, 0), bp = laneF(c.policy, 0);
                int ri = 0;
                if (ne > 1) { const float s1 = laneF(sc, 1), p1 = laneF(c.policy, 1); if (better(s1, p1, 1, bs, bp, ri)) { bs = s1; bp = p1; ri = 1; } }
                if (ne > 2) { const float s2 = laneF(sc, 2), p2 = laneF(c.policy, 2); if (better(s2, p2, 2, bs, bp, ri)) { bs = s2; bp = p2; ri = 2; } }
                cur.count = laneF(c.count, ri);
                cur.first_child = laneI(c.first_child, ri);
                cur.num_children = laneI(c.num_children, ri);
                cur.action = laneI(c.action, ri);
                cur.players = laneI(c.players, ri);
                node = fc + ri;
                note(depth, node, cur);
                if (lane == 0) {
                    path[depth] = node;
                    pact[depth] = cur.action;
                    if (hact) { hact[depth] = cur.action; }
                }
                ++depth;
                continue;
            }
            const bool two = ne > 64; // wave-uniform
            const bool has0 = lane < ne, has1 = lane + 64 < ne;
            NodeRec c0 = loadRec(recs + fc + (has0 ? lane : ne - 1)), c1 = c0;
            if (two) { c1 = loadRec(recs + fc + (has1 ? lane + 64 : ne - 1)); }
            const RcpPtr r0p = rcp + static_cast<int>(c0.count);
            const double r00 = r0p[0], r01 = r0p[1];
            double r10 = r00, r11 = r01;
            if (two) { const RcpPtr r1p = rcp + static_cast<int>(c1.count); r10 = r1p[0]; r11 = r1p[1]; }
            bool tiny0 = false, tiny1 = false;
            LevelEval e0 = evalChild(v, c0, cplayer, bias, sqrtN, r00, r01, &tiny0), e1 = e0;
            const bool vis0 = has0 && c0.count != 0.0f;
            bool vis1 = false;
            if (two) { e1 = evalChild(v, c1, cplayer, bias, sqrtN, r10, r11, &tiny1); vis1 = has1 && c1.count != 0.0f; }
            if (v.value_rescale || __ballot((vis0 && tiny0) || (vis1 && tiny1)) != 0) { // value rescaling, or a subnormal quotient (never seen in practice): the reference's own formula
                e0.q = normalizedMean(v, c0.reward, c0.mean, c0.count, cplayer, bsize, lo, hi);
                e1.q = normalizedMean(v, c1.reward, c1.mean, c1.count, cplayer, bsize, lo, hi);
            }
            // init Q: ordered f32 sum over the visited children in storage order (ref mcts.cpp:200-217)
            float sum_of_win = 0.0f, sum = 0.0f;
            unsigned long long m = __ballot(vis0);
            while (m) { const int j = __builtin_ctzll(m); m &= m - 1; sum_of_win += laneF(e0.q, j); sum += 1; }
            if (two) {
                m = __ballot(vis1);
                while (m) { const int j = __builtin_ctzll(m); m &= m - 1; sum_of_win += laneF(e1.q, j); sum += 1; }
            }
            const float init_q = v.atari_init_q ? (sum > 0 ? sum_of_win / sum : 1.0f) : (sum_of_win - 1) / (sum + 1);
            const float s0 = e0.score + (c0.count == 0.0f ? init_q : e0.q);
            const float s1 = e1.score + (c1.count == 0.0f ? init_q : e1.q);
            // per-lane best of its (at most) two children, then the wave arg-max: score, then prior, then the lower index (mcts.cpp:189-195)
            const bool take1 = has1 && better(s1, c1.policy, lane + 64, s0, c0.policy, lane);
            const NodeRec best = take1 ? c1 : c0;
            const int bi = take1 ? lane + 64 : lane;
            const unsigned ks = has0 ? orderedKey(take1 ? s1 : s0) : 0u;
            const unsigned smax = waveMaxU32(ks);
            const unsigned kp = (has0 && ks == smax) ? orderedKey(best.policy) : 0u;
            const unsigned pmax = waveMaxU32(kp);
            const unsigned ki = (has0 && ks == smax && kp == pmax) ? ~static_cast<unsigned>(bi) : 0u; // max of ~index = lowest index
            const int ri = static_cast<int>(~waveMaxU32(ki));
            const int owner = ri & 63;
            cur.count = laneF(best.count, owner);
            cur.first_child = laneI(best.first_child, owner);
            cur.num_children = laneI(best.num_children, owner);
            cur.action = laneI(best.action, owner);
            cur.players = laneI(best.players, owner);
            node = fc + ri;
            note(depth, node, cur);
            if (lane == 0) {
                path[depth] = node;
                pact[depth] = cur.action;
                if (hact) { hact[depth] = cur.action; }
            }
            ++depth;
            continue;
        }
        // ---- pass 1: init Q = ordered f32 sum over visited children (ref mcts.cpp:200-217); records stay in registers for A <= 128 ----
        NodeRec c0, c1;
        c0.count = 0; c1.count = 0;
        const bool has0 = lane < nc, has1 = lane + 64 < nc;
        if (has0) { c0 = loadRec(recs + fc + lane); }
        if (has1) { c1 = loadRec(recs + fc + lane + 64); }
        float sum_of_win = 0.0f, sum = 0.0f;
        float q0 = 0.0f, q1 = 0.0f;
        {
            const bool vis0 = has0 && c0.count != 0.0f;
            if (vis0) { q0 = normalizedMean(v, c0.reward, c0.mean, c0.count, cplayer, bsize, lo, hi); }
            unsigned long long m = __ballot(vis0);
            while (m) { const int j = __builtin_ctzll(m); m &= m - 1; sum_of_win += laneF(q0, j); sum += 1; }
            const bool vis1 = has1 && c1.count != 0.0f;
            if (vis1) { q1 = normalizedMean(v, c1.reward, c1.mean, c1.count, cplayer, bsize, lo, hi); }
            m = __ballot(vis1);
            while (m) { const int j = __builtin_ctzll(m); m &= m - 1; sum_of_win += laneF(q1, j); sum += 1; }
        }
        for (int cb = 128; cb < nc; cb += 64) { // wide nodes (A > 128): remaining chunks straight from memory
            const int i = cb + lane;
            float q = 0.0f;
            bool vis = false;
            if (i < nc) {
                const NodeRec c = loadRec(recs + fc + i);
                vis = c.count != 0.0f;
                if (vis) { q = normalizedMean(v, c.reward, c.mean, c.count, cplayer, bsize, lo, hi); }
            }
            unsigned long long m = __ballot(vis);
            while (m) { const int j = __builtin_ctzll(m); m &= m - 1; sum_of_win += laneF(q, j); sum += 1; }
        }
        const float init_q = v.atari_init_q ? (sum > 0 ? sum_of_win / sum : 1.0f) : (sum_of_win - 1) / (sum + 1);
        // ---- pass 2: PUCT score + arg-max (ref mcts.cpp:55-61,181-198) ----
        float bs = -FLT_MAX, bp = -FLT_MAX;
        int bi = INT_MAX;
        NodeRec best = c0;
        auto consider = [&](const NodeRec& c, float q, int i) {
            const float bpol = bias * c.policy;
            const float value_u = static_cast<float>((static_cast<double>(bpol) * sqrtN) / static_cast<double>(1 + c.count));
            const float value_q = (c.count == 0.0f) ? init_q : q;
            const float score = value_u + value_q;
            if (better(score, c.policy, i, bs, bp, bi)) { bs = score; bp = c.policy; bi = i; best = c; }
        };
        if (has0) { consider(c0, q0, lane); }
        if (has1) { consider(c1, q1, lane + 64); }
        for (int cb = 128; cb < nc; cb += 64) {
            const int i = cb + lane;
            if (i < nc) {
                const NodeRec c = loadRec(recs + fc + i);
                const float q = (c.count != 0.0f) ? normalizedMean(v, c.reward, c.mean, c.count, cplayer, bsize, lo, hi) : 0.0f;
                consider(c, q, i);
            }
        }
        // wave arg-max on the DPP network (row_shr 1/2/4/8, row_bcast 15/31: lane 63 ends up with the best triple) — no LDS traffic
        float rs = bs, rp = bp;
        int ri = bi;
        dppBest<0x111, 0xF>(rs, rp, ri);
        dppBest<0x112, 0xF>(rs, rp, ri);
        dppBest<0x114, 0xF>(rs, rp, ri);
        dppBest<0x118, 0xF>(rs, rp, ri);
        dppBest<0x142, 0xA>(rs, rp, ri);
        dppBest<0x143, 0xC>(rs, rp, ri);
        ri = laneI(ri, 63);
        // the lane that holds the winner broadcasts its record: that is the next level's header
        const int owner = __builtin_ctzll(__ballot(bi == ri));
        cur.count = laneF(best.count, owner);
        cur.first_child = laneI(best.first_child, owner);
        cur.num_children = laneI(best.num_children, owner);
        cur.action = laneI(best.action, owner);
        cur.players = laneI(best.players, owner);
        node = fc + ri;
        note(depth, node, cur);
        if (lane == 0) {
            path[depth] = node;
            pact[depth] = cur.action;
            if (hact) { hact[depth] = cur.action; }
        }
        ++depth;
    }
    if (lane == 0) {
        v.path_len[g] = depth;
        if (v.host_path_len) { v.host_path_len[g] = depth; }
        if (spec) { spec[0] = depth < kSpecCap ? depth : kSpecCap; }
    }
}

// One helper segment of the walk (wave `seg` = 1 .. kHelpSegs of the per-game simulation kernels, beside wave 0's selectBody): the kLv levels from level 1 + seg * kLv on of
// the path the previous walk took, evaluated in one pass of the same arithmetic as selectBody's speculation pass — every level's children block and header are checked
// against the records loaded here, the header of the segment's entry node comes from memory — into the segment's block.  Nothing but that block is written.
template <class RcpPtr>
__device__ __forceinline__ void selectSpecHelper(const PoolView& v, int g, int lane, int seg, int serial, RcpPtr rcp, SpecMem sm)
{
    if (!sm.w || v.atari_init_q || v.value_rescale || seg < 1 || seg > kHelpSegs) { return; }
    LdsI32* out = sm.w + kSpecHelp + (seg - 1) * kHelpSeg;
    const int way = __builtin_amdgcn_readfirstlane(sm.w[kSpecWays * kSpecWay + 6]) & (kSpecWays - 1);
    LdsI32* spec = sm.w + way * kSpecWay;
    const int plen = __builtin_amdgcn_readfirstlane(spec[0]);
    const int L0 = 1 + seg * kLv, depth = L0 + 1; // the entry node's level / the path index of the first chosen node
    const int max_depth = __builtin_amdgcn_readfirstlane(v.max_depth);
    int adv = 0, n0 = -1;
    int f_count = 0, f_fc = 0, f_nc = 0, f_act = 0, f_pl = 0, f_node = 0;
    if (L0 + 1 < plen && L0 + 1 < kSpecCap) {
        GNodeRec* recs = (GNodeRec*)(v.rec + size_t(g) * v.cap);
        MZ_GLOBAL const float* bias_tab = (MZ_GLOBAL const float*)v.bias_tab;
        MZ_GLOBAL const double* sqrt_tab = (MZ_GLOBAL const double*)v.sqrt_tab;
        int K = plen - L0 < kSpecCap - L0 ? plen - L0 : kSpecCap - L0;
        K = K < kLv ? K : kLv;
        const int k = lane / kGw, j = lane % kGw, Lk = L0 + k, base8 = lane - j;
        const bool inrange = k < K;
        n0 = __builtin_amdgcn_readfirstlane(spec[kSpecNode + L0]);
        const int nk = inrange ? spec[kSpecNode + Lk] : 0;
        const int fck = inrange ? spec[kSpecFc + Lk] : 0;
        const int nck = inrange ? spec[kSpecNc + Lk] : 0;
        const int pfc = (inrange && k >= 1) ? spec[kSpecFc + Lk - 1] : 0; // the block the predicted node lives in
        const int pnc = (inrange && k >= 1) ? spec[kSpecNc + Lk - 1] : 0;
        const int nl = nck < kGw ? nck : kGw;
        const bool ld = inrange && j < nl && fck >= 0 && fck + j < v.cap;
        const NodeRec c = loadRec(recs + (ld ? fck + j : 0));
        const NodeRec h0 = loadRec(recs + ((n0 >= 0 && n0 < v.cap) ? n0 : 0)); // the entry node's own record: its header is not among the records of a level before it
        const int idx = nk - pfc;
        const bool idx_ok = idx >= 0 && idx < kGw && idx < pnc;
        const int srcl = idx_ok ? base8 - kGw + idx : lane;
        float hcount = __shfl(c.count, srcl);
        int hfc = __shfl(c.first_child, srcl), hnc = __shfl(c.num_children, srcl), hpl = __shfl(c.players, srcl);
        if (k == 0) { hcount = h0.count; hfc = h0.first_child; hnc = h0.num_children; hpl = h0.players; }
        bool okk = inrange && ld == (j < nl) && nck > 0 && n0 >= 0 && n0 < v.cap && (k == 0 || idx_ok) && hfc == fck && hnc == nck && hcount >= 1.0f;
        int Nk = static_cast<int>(hcount - 1);
        Nk = okk ? Nk : 0;
        const float biask = sm.bias ? sm.bias[Nk] : bias_tab[Nk];
        const double sqrtNk = sm.sqrt ? sm.sqrt[Nk] : sqrt_tab[Nk];
        const unsigned visn = static_cast<unsigned>(hpl) >> 16;
        const int nek = (visn == 0xFFFFu) ? nck : (nck < static_cast<int>(visn) + 1 ? nck : static_cast<int>(visn) + 1);
        okk = okk && nek <= kGw;
        const int cplk = (hpl >> 8) & 0xFF;
        const bool has = okk && j < nek;
        const RcpPtr rp = rcp + (has ? static_cast<int>(c.count) : 0);
        bool tiny = false;
        const LevelEval e = evalChild(v, c, cplk, biask, sqrtNk, rp[0], rp[1], &tiny);
        const bool vis = has && c.count != 0.0f;
        if (__ballot(vis && tiny) == 0) { // (a subnormal quotient needs the reference's division: left to the plain walk)
            int mx = 0;
#pragma unroll
            for (int t = 1; t <= kGw; ++t) { if (__ballot(okk && nek >= t) != 0) { mx = t; } }
            const float qm = vis ? e.q : 0.0f, vm1 = vis ? 1.0f : 0.0f;
            float sum_of_win = 0.0f, sum = 0.0f;
            for (int i = 0; i < mx; ++i) {
                const float qi = __shfl(qm, base8 + i), vi = __shfl(vm1, base8 + i);
                sum_of_win = sum_of_win + qi;
                sum = sum + vi;
            }
            const float init_q = (sum_of_win - 1) / (sum + 1);
            const float sc = e.score + (c.count == 0.0f ? init_q : e.q);
            float bs = 0.0f, bp = 0.0f;
            int bi = 0;
            for (int i = 0; i < mx; ++i) {
                const float si = __shfl(sc, base8 + i), pi = __shfl(c.policy, base8 + i);
                if (i < nek && (i == 0 || better(si, pi, i, bs, bp, bi))) { bs = si; bp = pi; bi = i; }
            }
            const int chl = base8 + bi; // the lane that holds the level's chosen child
            const float ch_count = __shfl(c.count, chl);
            const int ch_fc = __shfl(c.first_child, chl), ch_nc = __shfl(c.num_children, chl), ch_act = __shfl(c.action, chl), ch_pl = __shfl(c.players, chl);
            const int chosen = fck + bi;
            const int prev_chosen = __shfl(chosen, lane >= kGw ? lane - kGw : lane), prev_nc = __shfl(ch_nc, lane >= kGw ? lane - kGw : lane);
            const bool link = okk && (k == 0 || (prev_chosen == nk && prev_nc != 0)) && depth + k < max_depth;
            const unsigned long long lm = __ballot(j == 0 && !link);
            const int adv_all = lm ? static_cast<int>(__builtin_ctzll(lm)) / kGw : kLv;
            adv = adv_all < K ? adv_all : K;
            if (adv > 0) {
                if (j == 0 && k < adv) {
                    out[kHelpHdr + k] = chosen;
                    out[kHelpHdr + kLv + k] = ch_act;
                    out[kHelpHdr + 2 * kLv + k] = ch_fc;
                    out[kHelpHdr + 3 * kLv + k] = ch_nc;
                }
                const int l = kGw * (adv - 1);
                f_count = __builtin_amdgcn_readlane(__float_as_int(ch_count), l);
                f_fc = laneI(ch_fc, l); f_nc = laneI(ch_nc, l); f_act = laneI(ch_act, l); f_pl = laneI(ch_pl, l); f_node = laneI(chosen, l);
            }
        }
    }
    if (lane == 0) {
        out[2] = n0; out[3] = adv; out[4] = f_count; out[5] = f_fc; out[6] = f_nc; out[7] = f_act; out[8] = f_pl; out[9] = f_node;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { out[0] = serial; }
}

// what expandBackupBody otherwise reads from the per-game staging arrays, for a caller that holds it in registers
struct ExpandGiven {
    int k, player;
    float value, reward;
};

// `lds` = 2 * bound_cap words of LDS (value-bound multiset: keys then counts; only touched with value_rescale)
__device__ __forceinline__ void expandBackupBody(const PoolView& v, const int* __restrict__ cand_count, const int* __restrict__ cand_action,
                                                 const float* __restrict__ cand_policy, const float* __restrict__ cand_logit,
                                                 const int* __restrict__ cand_player, const float* __restrict__ value_in,
                                                 const float* __restrict__ reward_in, int hslot, int* __restrict__ err, int g, int lane,
                                                 float* __restrict__ lds, int part = 0, const ExpandGiven* given = nullptr)
{
    // part 0: expand, then backup (one wave).  The two touch different words of the tree (expand: the new children, the leaf's child block and
    // slot; backup: mean / count of the path's nodes and the parent's visited-prefix counter), so without value rescaling two waves can run
    // them side by side: part 1 = expand only, part 2 = backup only (cand_count is not read).
    const size_t base = size_t(g) * v.cap;
    const int len = v.path_len[g];
    if (len <= 0) { return; }
    const int* path = v.path + size_t(g) * v.max_depth;
    const int leaf = path[len - 1];
    const int k = part == 2 ? 0 : (given ? given->k : cand_count[g]);
    // ---- expand (ref mcts.cpp:151-164, tree.h:71-77) ----
    if (k > 0) {
        const int fc = v.num_nodes[g];
        if (fc + k > v.cap) {
            if (lane == 0) { atomicExch(err, MZ_ERR_CAPACITY); }
            return;
        }
        const int pl = given ? given->player : cand_player[g];
        bool unsorted = false; // select's visited-prefix shortcut needs the priors in descending order (the actor always sorts them)
        for (int i = lane; i < k; i += 64) {
            const size_t n = base + fc + i, c = size_t(g) * v.A + i;
            if (i > 0 && cand_policy[c - 1] < cand_policy[c]) { unsorted = true; }
            NodeRec r;
            r.count = 0; r.mean = 0; r.policy = cand_policy[c]; r.reward = 0;
            r.first_child = -1; r.num_children = 0; r.action = cand_action[c]; r.players = pl;
            reinterpret_cast<float4*>(v.rec + n)[0] = make_float4(r.count, r.mean, r.policy, r.reward);
            reinterpret_cast<int4*>(v.rec + n)[1] = make_int4(r.first_child, r.num_children, r.action, r.players);
            v.logit[n] = cand_logit[c];
            v.noise[n] = 0; v.value[n] = 0; v.hslot[n] = -1;
        }
        const bool any_unsorted = __ballot(unsorted) != 0;
        if (lane == 0) {
            NodeRec* l = v.rec + base + leaf;
            l->first_child = fc;
            l->num_children = k;
            // bits 16..31: number of visited children (0xFFFF = unknown order: select looks at every child)
            l->players = (l->players & 0xFF) | (pl << 8) | (any_unsorted ? static_cast<int>(0xFFFF0000u) : 0);
            v.num_nodes[g] = fc + k;
        }
    }
    if (part != 2 && lane == 0 && hslot >= 0) { v.hslot[base + leaf] = hslot; }
    MZ_LPROF(10);
    if (part == 1) { return; }
    // ---- backup (ref mcts.cpp:166-179) ----
    if (!v.value_rescale) {
        // The only leaf -> root dependence is `updated = r + gamma * updated`, which needs the rewards but not the means: the
        // path's records are loaded by 64 lanes at once (one memory round trip per 64 levels instead of one per level — the
        // deepest of the 256 paths sets the kernel time), the chain runs over registers, then every lane updates its own node.
        const float val = given ? given->value : value_in[g], rew = given ? given->reward : reward_in[g];
        if (lane == 0) {
            v.value[base + leaf] = val;
            v.rec[base + leaf].reward = rew;
        }
        float updated = val;
        for (int kb = 0; kb < len; kb += 64) {
            const int k = kb + lane; // k-th node from the leaf
            const bool act = k < len;
            NodeRec* n = v.rec + base + (act ? path[len - 1 - k] : 0);
            float mean = 0.0f, cnt = 0.0f, r = 0.0f;
            if (act) {
                mean = n->mean;
                cnt = n->count;
                r = (k == 0) ? rew : n->reward;
            }
            if (kb == 0 && len >= 2 && laneF(cnt, 0) == 0.0f && lane == 1 && (static_cast<unsigned>(n->players) >> 16) != 0xFFFFu) { n->players += 1 << 16; } // the leaf's parent: one more visited child (select's prefix)
            float mine = 0.0f;
            const int m = len - kb < 64 ? len - kb : 64;
            for (int j = 0; j < m; ++j) {
                if (lane == j) { mine = updated; }
                updated = laneF(r, j) + v.gamma * updated;
            }
            if (act) { // MCTSNode::add(value, 1.0f) (ref mcts.cpp:20-28)
                cnt += 1.0f;
                mean += 1.0f * (mine - mean) / cnt;
                n->mean = mean;
                n->count = cnt;
            }
        }
        MZ_LPROF(11);
        return;
    }
    // with value rescaling the value-bound multiset (std::map<float,int> of the reference, kept as an unordered array) is updated node by
    // node: a serial chain over the path, but every step is wave-cooperative — the two key searches are ballots over 64 entries at a
    // time (a 50-simulation Atari search holds hundreds of distinct values), the copy in / out of LDS and the final min / max too
    float* bkey = lds;                                     // value-bound multiset, LDS copy
    int* bcnt = reinterpret_cast<int*>(lds + v.bound_cap);
    int bsize = v.bound_size[g];
    // Up to 64 entries (num_simulation <= 61: BASELINE configs[4] has 53) the multiset lives in REGISTERS, entry j in lane j: a key search is one ballot, an
    // update a readlane and a predicated move — no LDS round trips and no fences between the steps of the chain (3.8 -> 1.x us per backup)
    const bool in_regs = v.bound_cap <= 64;
    float rkey = 0.0f;
    int rcnt = 0;
    if (in_regs) {
        if (lane < v.bound_cap) { rkey = v.bound_key[size_t(g) * v.bound_cap + lane]; rcnt = v.bound_cnt[size_t(g) * v.bound_cap + lane]; }
    } else {
        // (all bound_cap entries, not only the bsize live ones: the loads then do not wait for bsize; the others are never looked at)
        for (int j = lane; j < v.bound_cap; j += 64) { bkey[j] = v.bound_key[size_t(g) * v.bound_cap + j]; bcnt[j] = v.bound_cnt[size_t(g) * v.bound_cap + j]; }
    }
    const float val = given ? given->value : value_in[g], rew = given ? given->reward : reward_in[g];
    // The records of the path's nodes are fetched by 64 lanes at once (lane k = the k-th node from the leaf), together with the multiset: ONE round trip in front
    // of the chain.  Nothing is stored to global memory before the chain is through — the chain's steps are separated by fences (the multiset lives in LDS, but a
    // release fence also waits for every global store in flight: with the nodes' new mean / count stored inside the loop each level paid a store round trip,
    // 3.9 us per backup on BASELINE configs[4]); lane k keeps the new statistics of its node and stores them at the end.
    float pmean = 0.0f, pcnt = 0.0f, prew = 0.0f;
    int ppl = 0;
    if (lane < len) {
        const NodeRec* pn = v.rec + base + path[len - 1 - lane];
        pmean = pn->mean; pcnt = pn->count; prew = pn->reward; ppl = pn->players;
    }
    float my_mean = 0.0f, my_cnt = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    auto find = [&](float key) { // index of the entry with this key (f32 ==: -0 and +0 are one key), -1 if none; wave-uniform
        for (int b0 = 0; b0 < bsize; b0 += 64) {
            const int j = b0 + lane;
            const unsigned long long m = __ballot(j < bsize && bkey[j] == key);
            if (m) { return b0 + static_cast<int>(__builtin_ctzll(m)); }
        }
        return -1;
    };
    auto sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    float updated = val;
    for (int i = len - 1; i >= 0; --i) {
        NodeRec* n = v.rec + base + path[i];
        const int kk = len - 1 - i; // wave-uniform
        const float r = (i == len - 1) ? rew : (kk < 64 ? laneF(prew, kk) : n->reward);
        float mean = kk < 64 ? laneF(pmean, kk) : n->mean, cnt = kk < 64 ? laneF(pcnt, kk) : n->count;
        const float old_mean = r + v.gamma * mean;
        // MCTSNode::add(value, 1.0f) (ref mcts.cpp:20-28); count + 1 <= 0 cannot happen for count >= 0
        cnt += 1.0f;
        mean += 1.0f * (updated - mean) / cnt;
        if (kk < 64) {
            if (lane == kk) { my_mean = mean; my_cnt = cnt; }
        } else if (lane == 0) { // (levels beyond the 64 the lanes hold)
            n->mean = mean;
            n->count = cnt;
        }
        if (in_regs) { // updateTreeValueBound(old, new) (ref mcts.cpp:219-228) on the register copy: the same steps as below
            const float new_mean = r + v.gamma * mean;
            const unsigned long long mo = __ballot(lane < bsize && rkey == old_mean);
            if (mo) {
                const int jo = static_cast<int>(__builtin_ctzll(mo));
                const int c = __builtin_amdgcn_readlane(rcnt, jo) - 1;
                if (c == 0) {
                    --bsize;
                    const float lk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rkey), bsize));
                    const int lc = __builtin_amdgcn_readlane(rcnt, bsize);
                    if (lane == jo) { rkey = lk; rcnt = lc; }
                } else if (lane == jo) {
                    rcnt = c;
                }
            }
            const unsigned long long mn = __ballot(lane < bsize && rkey == new_mean);
            if (mn) {
                if (lane == static_cast<int>(__builtin_ctzll(mn))) { ++rcnt; }
            } else if (bsize < v.bound_cap) {
                if (lane == bsize) { rkey = new_mean; rcnt = 1; }
                ++bsize;
            }
        } else { // updateTreeValueBound(old, new) (ref mcts.cpp:219-228)
            const float new_mean = r + v.gamma * mean;
            const int jo = find(old_mean);
            if (jo >= 0) {
                const int c = bcnt[jo] - 1;
                if (c == 0) { --bsize; }
                if (lane == 0) {
                    if (c == 0) { bkey[jo] = bkey[bsize]; bcnt[jo] = bcnt[bsize]; } else { bcnt[jo] = c; }
                }
                sync();
            }
            const int jn = find(new_mean);
            if (jn >= 0) {
                if (lane == 0) { ++bcnt[jn]; }
            } else if (bsize < v.bound_cap) {
                if (lane == 0) { bkey[bsize] = new_mean; bcnt[bsize] = 1; }
                ++bsize;
            }
            sync();
        }
        updated = r + v.gamma * updated;
    }
    if (lane < len && lane < 64) { // the nodes' new statistics; the leaf's value and reward; one more visited child of the leaf's parent (select's prefix)
        NodeRec* n = v.rec + base + path[len - 1 - lane];
        n->mean = my_mean;
        n->count = my_cnt;
        if (lane == 0) {
            v.value[base + leaf] = val;
            n->reward = rew;
        }
        if (lane == 1 && laneF(pcnt, 0) == 0.0f && (static_cast<unsigned>(ppl) >> 16) != 0xFFFFu) { n->players = ppl + (1 << 16); }
    }
    {
        float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
        if (in_regs) {
            if (lane < bsize) {
                v.bound_key[size_t(g) * v.bound_cap + lane] = rkey;
                v.bound_cnt[size_t(g) * v.bound_cap + lane] = rcnt;
                lo = rkey; hi = rkey;
            }
        } else
        for (int j = lane; j < bsize; j += 64) {
            v.bound_key[size_t(g) * v.bound_cap + j] = bkey[j];
            v.bound_cnt[size_t(g) * v.bound_cap + j] = bcnt[j];
            lo = bkey[j] < lo ? bkey[j] : lo;
            hi = bkey[j] > hi ? bkey[j] : hi;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lane == 0) {
            v.bound_size[g] = bsize;
            v.bound_lo[g] = bsize > 0 ? lo : 0.0f;
            v.bound_hi[g] = bsize > 0 ? hi : 0.0f;
        }
    }
}

} // namespace mz
