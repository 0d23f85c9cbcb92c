// Device-resident Go leaf environment + candidate lists (see go_dev.h).  One wave64 per game; point p = i * 64 + lane is
// cell i of lane `lane`, so a ballot over cell i is word i of a bitboard.  All group bookkeeping is O(1) passes over the
// cells: no flood fill, no replay.  Integer / bit work only (bit-exact against the host engine by construction of the
// same rules; checked by tests/test_gpu_godev.py playouts and end to end by tests/test_gpu_worker.py).
#include "go_dev.h"
#include "sort_emul.h"
#include <cstring>
#include <vector>

namespace mz {

namespace {

struct Cand { int action; float policy, logit; };
struct CandGreater { __host__ __device__ bool operator()(const Cand& l, const Cand& r) const { return l.policy > r.policy; } };
using CandSort = StdSortEmul<Cand, CandGreater>;
constexpr size_t kSortStackBytes = 3 * CandSort::kStack * sizeof(int);

__device__ inline uint64_t shflXor64(uint64_t v, int o)
{
    const unsigned lo = __shfl_xor(static_cast<unsigned>(v), o), hi = __shfl_xor(static_cast<unsigned>(v >> 32), o);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ inline uint64_t waveXor64(uint64_t v)
{
    for (int o = 32; o > 0; o >>= 1) { v ^= shflXor64(v, o); }
    return v;
}
__device__ inline uint64_t normH(uint64_t h) { return h ? h : 1; }
__device__ inline int rotOf(const RotPack& r, int g) { return (r.w[g / 10] >> (3 * (g % 10))) & 7; }

// snapshot -> slot 0 of the game's position slab
__global__ __launch_bounds__(64) void go_root_kernel(GoDevView v)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    const GoRootSnapshot& S = v.snap[g];
    const size_t sb = size_t(g) * v.slots;
    if (lane < 2 * v.W) { v.stones[sb * 2 * v.W + lane] = S.stones[lane / v.W][lane % v.W]; }
    if (lane == 0) {
        v.hash[sb] = S.hash;
        v.meta[sb * 2] = S.nmoves;
        v.meta[sb * 2 + 1] = S.passes;
    }
    for (int p = lane; p < v.P; p += 64) { v.lab[sb * v.Ppad + p] = S.lab[p]; }
}

template <int CPL>
__global__ __launch_bounds__(64) void go_leaf_kernel(GoDevView v, PoolView pv, RotPack rp, int slot)
{
    extern __shared__ uint64_t smem[];
    const int g = blockIdx.x, lane = threadIdx.x;
    const int P = v.P, n = v.n, W = v.W, Ppad = v.Ppad, MD = pv.max_depth;
    uint64_t* gh = smem;                                   // [Ppad] XOR of the keys of a group, by group id
    uint64_t* ph = gh + Ppad;                              // [MD]   (normalised) hashes of the positions along the path
    uint64_t* hb = ph + MD;                                // [8][2][W] stones k moves before the leaf
    uint64_t* cur = hb + 16 * W;                           // [2][W] stones at the leaf
    int* libs = reinterpret_cast<int*>(cur + 2 * W);       // [Ppad] liberties of a group, by group id
    uint16_t* lab = reinterpret_cast<uint16_t*>(libs + Ppad); // [Ppad] group id per point
    uint8_t* col = reinterpret_cast<uint8_t*>(lab + Ppad);  // [Ppad] 0 empty, 1 black, 2 white, 3 off board

    const int len = pv.path_len[g];
    const int* path = pv.path + size_t(g) * MD;
    const int* pact = pv.path_action + size_t(g) * MD;
    const int depth = len - 1;
    const GoRootSnapshot& S = v.snap[g];
    const int root_turn = S.turn, root_hist_len = S.hist_len;
    const size_t sb = size_t(g) * v.slots;
    const int* hs = pv.hslot + size_t(g) * pv.cap;
    const int src = depth == 0 ? 0 : hs[path[len - 2]];

    int c[CPL], l[CPL];
    short nb[CPL][4];
    uint64_t sbw[CPL], sww[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int p = i * 64 + lane;
        sbw[i] = v.stones[((sb + src) * 2 + 0) * W + i];
        sww[i] = v.stones[((sb + src) * 2 + 1) * W + i];
        c[i] = 3;
        l[i] = 0;
        nb[i][0] = nb[i][1] = nb[i][2] = nb[i][3] = -1;
        if (p < P) {
            c[i] = ((sbw[i] >> lane) & 1) ? 1 : (((sww[i] >> lane) & 1) ? 2 : 0);
            l[i] = v.lab[(sb + src) * Ppad + p];
            const int x = p % n, y = p / n;
            if (y + 1 < n) { nb[i][0] = static_cast<short>(p + n); }
            if (x + 1 < n) { nb[i][1] = static_cast<short>(p + 1); }
            if (y > 0) { nb[i][2] = static_cast<short>(p - n); }
            if (x > 0) { nb[i][3] = static_cast<short>(p - 1); }
        }
        col[p] = static_cast<uint8_t>(c[i]);
        lab[p] = static_cast<uint16_t>(l[i]);
    }
    uint64_t hash = v.hash[sb + src];
    int nmoves = v.meta[(sb + src) * 2], passes = v.meta[(sb + src) * 2 + 1];
    __syncthreads();
    const int t = (depth & 1) ? 3 - root_turn : root_turn; // the player to move at the leaf

    if (depth >= 1) { // leaf = parent + one move (ref go.cpp:132-190, observable effects only)
        const int a = pact[len - 1], m = 3 - t;
        ++nmoves;
        if (a >= P) {
            passes = passes + 1 > 2 ? 2 : passes + 1;
        } else {
            passes = 0;
            const int ax = a % n, ay = a / n;
            const int an[4] = {ay + 1 < n ? a + n : -1, ax + 1 < n ? a + 1 : -1, ay > 0 ? a - n : -1, ax > 0 ? a - 1 : -1};
            int own[4], en[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                own[k] = -1;
                en[k] = -1;
                if (an[k] >= 0) {
                    const int cq = col[an[k]];
                    if (cq == m) { own[k] = lab[an[k]]; }
                    else if (cq == 3 - m) { en[k] = lab[an[k]]; }
                }
            }
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                for (int j = 0; j < k; ++j) { if (en[j] == en[k]) { en[k] = -1; } }
            }
            __syncthreads();
            // place the stone; the own groups it touches become one group whose id is the new point (unused as an id: it was empty)
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int p = i * 64 + lane;
                if (p == a) { c[i] = m; l[i] = a; }
                else if (c[i] == m && (l[i] == own[0] || l[i] == own[1] || l[i] == own[2] || l[i] == own[3])) { l[i] = a; }
                col[p] = static_cast<uint8_t>(c[i]);
                lab[p] = static_cast<uint16_t>(l[i]);
            }
            __syncthreads();
            // which of the adjacent enemy groups still have a liberty
            unsigned flags = 0;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                if (c[i] != 0) { continue; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int q = nb[i][k];
                    if (q >= 0 && col[q] == 3 - m) {
                        const int lq = lab[q];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { if (lq == en[j]) { flags |= 1u << j; } }
                    }
                }
            }
            bool cap[4], any_cap = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cap[j] = en[j] >= 0 && __ballot((flags >> j) & 1) == 0;
                any_cap |= cap[j];
            }
            uint64_t hx = 0;
            if (any_cap) {
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    const int p = i * 64 + lane;
                    if (c[i] == 3 - m && ((cap[0] && l[i] == en[0]) || (cap[1] && l[i] == en[1]) || (cap[2] && l[i] == en[2]) || (cap[3] && l[i] == en[3]))) {
                        c[i] = 0;
                        col[p] = 0;
                        hx ^= v.key[size_t(2 - m) * P + p];
                    }
                }
                hx = waveXor64(hx);
            }
            hash ^= v.key[size_t(m - 1) * P + a] ^ hx;
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = i * 64 + lane;
            sbw[i] = __ballot(c[i] == 1);
            sww[i] = __ballot(c[i] == 2);
            if (lane == 0) {
                v.stones[((sb + slot) * 2 + 0) * W + i] = sbw[i];
                v.stones[((sb + slot) * 2 + 1) * W + i] = sww[i];
            }
            if (p < P) { v.lab[(sb + slot) * Ppad + p] = static_cast<uint16_t>(l[i]); }
        }
        if (lane == 0) {
            v.hash[sb + slot] = hash;
            v.meta[(sb + slot) * 2] = nmoves;
            v.meta[(sb + slot) * 2 + 1] = passes;
        }
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        if (lane == 0) { cur[i] = sbw[i]; cur[W + i] = sww[i]; }
    }
    const bool terminal = passes >= 2 || nmoves > 2 * P; // ref go.cpp:246-257
    // ---- hashes along the path (d = 1 .. depth; the root and everything before it is in the root's table) ----
    for (int d = 1 + lane; d <= depth; d += 64) { ph[d - 1] = normH(d == depth ? hash : v.hash[sb + hs[path[d]]]); }
    // ---- group liberties / key sums at the leaf ----
#pragma unroll
    for (int i = 0; i < CPL; ++i) { libs[i * 64 + lane] = 0; gh[i * 64 + lane] = 0; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        if (c[i] != 0) { continue; }
        int seen_l[4], ns = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = nb[i][k];
            if (q < 0 || col[q] == 0) { continue; }
            const int lq = lab[q];
            bool dup = false;
            for (int j = 0; j < ns; ++j) { dup |= seen_l[j] == lq; }
            if (!dup) { seen_l[ns++] = lq; atomicAdd(&libs[lq], 1); }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int p = i * 64 + lane;
        if (c[i] == 3 - t && libs[l[i]] == 1) { atomicXor(reinterpret_cast<unsigned long long*>(&gh[l[i]]), static_cast<unsigned long long>(v.key[size_t(2 - t) * P + p])); }
    }
    __syncthreads();
    // ---- legal mask for the player to move (ref go.cpp:208-244): not occupied, not suicide, not a positional-superko repeat ----
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        bool bit = false;
        if (c[i] == 0 && !terminal) {
            const int p = i * 64 + lane;
            bool ok = false;
            uint64_t nh = hash ^ v.key[size_t(t - 1) * P + p];
            int capl[4], ncap = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = nb[i][k];
                if (q < 0) { continue; }
                const int cq = col[q];
                if (cq == 0) { ok = true; continue; }
                const int lq = lab[q];
                if (cq == t) {
                    if (libs[lq] > 1) { ok = true; }
                } else if (libs[lq] == 1) { // an enemy group in atari is captured: each group once
                    bool dup = false;
                    for (int j = 0; j < ncap; ++j) { dup |= capl[j] == lq; }
                    if (!dup) { capl[ncap++] = lq; nh ^= gh[lq]; }
                    ok = true;
                }
            }
            if (ok) {
                const uint64_t h = normH(nh);
                bool rep = false;
                for (uint32_t s = static_cast<uint32_t>(h) & (kGoSeenCap - 1);; s = (s + 1) & (kGoSeenCap - 1)) {
                    const uint64_t e = S.seen[s];
                    if (e == 0) { break; }
                    if (e == h) { rep = true; break; }
                }
                for (int d = 0; d < depth && !rep; ++d) { rep = ph[d] == h; }
                bit = !rep;
            }
        }
        uint64_t w = __ballot(bit);
        if (i == (P >> 6)) { w |= 1ull << (P & 63); } // pass is always legal
        if (lane == 0) { v.legal[size_t(g) * v.LW + i] = w; }
    }
    if (v.LW > CPL && lane == 0) { v.legal[size_t(g) * v.LW + CPL] = (P >> 6) == CPL ? 1ull << (P & 63) : 0; } // P a multiple of 64
    // ---- feature planes (ref go.cpp:280-308): planes 2k / 2k+1 = own / opponent stones k moves ago, 16 / 17 = black / white to move ----
    const int avail = root_hist_len + depth;
    for (int idx = lane; idx < 16 * W; idx += 64) {
        const int k = idx / (2 * W), cw = idx % (2 * W);
        uint64_t val = 0;
        if (k < avail) {
            if (k == 0) { val = cur[cw]; }
            else if (k < depth) { val = v.stones[(sb + hs[path[len - 1 - k]]) * 2 * W + cw]; }
            else { val = S.hist[(root_hist_len - 1 - (k - depth)) & 7][cw / W][cw % W]; }
        }
        hb[idx] = val;
    }
    __syncthreads();
    {
        const uint16_t* map = v.inv + size_t(rotOf(rp, g)) * P;
        uint32_t* out = v.feat + size_t(g) * 18 * v.W32;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = i * 64 + lane;
            const int q = p < P ? map[p] : 0;
            uint64_t mine = 0; // lane ch keeps plane ch's word
            for (int ch = 0; ch < 16; ++ch) {
                const int k = ch >> 1, color = (ch & 1) == 0 ? t - 1 : 2 - t;
                const bool bit = p < P && ((hb[(k * 2 + color) * W + (q >> 6)] >> (q & 63)) & 1);
                const uint64_t bal = __ballot(bit);
                if (lane == ch) { mine = bal; }
            }
            const uint64_t ones = __ballot(p < P);
            if (lane == 16) { mine = t == 1 ? ones : 0; }
            if (lane == 17) { mine = t == 2 ? ones : 0; }
            if (lane < 18) {
                if (2 * i < v.W32) { out[lane * v.W32 + 2 * i] = static_cast<uint32_t>(mine); }
                if (2 * i + 1 < v.W32) { out[lane * v.W32 + 2 * i + 1] = static_cast<uint32_t>(mine >> 32); }
            }
        }
    }
    // ---- terminal: Tromp-Taylor area score + komi (ref go.cpp:259-278,703-723) ----
    float eval = 0.0f;
    if (terminal) {
        __syncthreads();
        uint16_t* rl = lab; // region id of the empty points: the smallest point of the region
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = i * 64 + lane;
            rl[p] = static_cast<uint16_t>(p);
            libs[p] = 0;
            gh[p] = 0;
        }
        __syncthreads();
        while (true) {
            bool changed = false;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                if (c[i] != 0) { continue; }
                const int p = i * 64 + lane;
                int mn = rl[p];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int q = nb[i][k];
                    if (q >= 0 && col[q] == 0 && rl[q] < mn) { mn = rl[q]; }
                }
                if (mn < rl[p]) { rl[p] = static_cast<uint16_t>(mn); changed = true; }
            }
            __syncthreads();
            if (__ballot(changed) == 0) { break; }
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            if (c[i] != 0) { continue; }
            const int p = i * 64 + lane, r = rl[p];
            atomicAdd(&libs[r], 1);
            unsigned long long border = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = nb[i][k];
                if (q >= 0 && (col[q] == 1 || col[q] == 2)) { border |= col[q]; }
            }
            if (border) { atomicOr(reinterpret_cast<unsigned long long*>(&gh[r]), border); }
        }
        __syncthreads();
        float t1 = 0.0f, t2 = 0.0f;
        for (int i = 0; i < W; ++i) { t1 += static_cast<float>(__popcll(cur[i])); t2 += static_cast<float>(__popcll(cur[W + i])); }
        t2 += v.komi;
#pragma unroll
        for (int i = 0; i < CPL; ++i) { // regions in ascending order of their first point, like the host's scan
            const int p = i * 64 + lane;
            uint64_t roots = __ballot(c[i] == 0 && rl[p] == p);
            while (roots) {
                const int r = i * 64 + __builtin_ctzll(roots);
                roots &= roots - 1;
                const uint64_t border = gh[r];
                const float size = static_cast<float>(libs[r]);
                if ((border & 2) == 0) { t1 += size; }
                else if ((border & 1) == 0) { t2 += size; }
            }
        }
        eval = t1 > t2 ? 1.0f : (t1 < t2 ? -1.0f : 0.0f);
    }
    if (lane == 0) {
        v.leaf_player[g] = t;
        v.terminal[g] = terminal ? 1 : 0;
        v.eval[g] = eval;
    }
}

// order `k` candidates in cs[] like the reference's std::sort(policy descending): result in out[]
__device__ void orderCandidates(Cand* cs, Cand* out, int* stack, int k, int lane, int* err)
{
    bool tie = false;
    for (int i = lane; i < k; i += 64) {
        const float pi = cs[i].policy;
        int rank = 0;
        for (int j = 0; j < k; ++j) {
            const float pj = cs[j].policy;
            rank += (pj > pi) || (pj == pi && j < i);
            tie |= (pj == pi && j != i);
        }
        out[rank] = cs[i];
    }
    __syncthreads();
    if (k > 16 && __ballot(tie) != 0) { // ties among > 16 elements: only the exact introsort gives the reference's order
        if (lane == 0) {
            StdSortEmul<Cand, CandGreater> s{cs, CandGreater()};
            if (!s.sort(k, stack) && err) { atomicExch(err, MZ_ERR_CAPACITY); }
        }
        __syncthreads();
        for (int i = lane; i < k; i += 64) { out[i] = cs[i]; }
        __syncthreads();
    }
}

// AlphaZero candidates of a leaf (ref zero_actor.cpp:215-245): legal actions in action order, policy / logit looked up through
// the rotation, sorted by policy; a terminal leaf has no children and its value is the game result (zero_actor.cpp:85)
__global__ __launch_bounds__(64) void az_cand_kernel(GoDevView v, const float* __restrict__ policy, const float* __restrict__ logit,
                                                     const float* __restrict__ value, RotPack rp, int* __restrict__ cand_count,
                                                     int* __restrict__ cand_action, float* __restrict__ cand_policy, float* __restrict__ cand_logit,
                                                     int* __restrict__ cand_player, float* __restrict__ value_out, float* __restrict__ reward_out,
                                                     int* __restrict__ err)
{
    extern __shared__ uint64_t smem[];
    Cand* cs = reinterpret_cast<Cand*>(smem);
    Cand* out = cs + v.A;
    int* stack = reinterpret_cast<int*>(out + v.A);
    const int g = blockIdx.x, lane = threadIdx.x, A = v.A;
    const bool terminal = v.terminal[g] != 0;
    int k = 0;
    if (!terminal) {
        const uint16_t* fwd = v.fwd + size_t(rotOf(rp, g)) * A;
        for (int base = 0; base < A; base += 64) {
            const int a = base + lane;
            const bool leg = a < A && ((v.legal[size_t(g) * v.LW + (a >> 6)] >> (a & 63)) & 1);
            const uint64_t m = __ballot(leg);
            if (leg) {
                const int pos = k + __popcll(m & ((1ull << lane) - 1));
                const int f = fwd[a];
                cs[pos] = Cand{a, policy[size_t(g) * A + f], logit[size_t(g) * A + f]};
            }
            k += __popcll(m);
        }
        __syncthreads();
        orderCandidates(cs, out, stack, k, lane, err);
        for (int i = lane; i < k; i += 64) {
            cand_action[size_t(g) * A + i] = out[i].action;
            cand_policy[size_t(g) * A + i] = out[i].policy;
            cand_logit[size_t(g) * A + i] = out[i].logit;
        }
    }
    if (lane == 0) {
        cand_count[g] = k;
        cand_player[g] = v.leaf_player[g];
        value_out[g] = terminal ? v.eval[g] : value[g];
        reward_out[g] = 0.0f;
    }
}

__global__ __launch_bounds__(64) void sort_test_kernel(const float* __restrict__ policy, int n, int* __restrict__ order, int* __restrict__ err)
{
    extern __shared__ uint64_t smem[];
    Cand* cs = reinterpret_cast<Cand*>(smem);
    Cand* out = cs + n;
    int* stack = reinterpret_cast<int*>(out + n);
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) { cs[i] = Cand{i, policy[i], 0.0f}; }
    __syncthreads();
    orderCandidates(cs, out, stack, n, lane, err);
    for (int i = lane; i < n; i += 64) { order[i] = out[i].action; }
}

} // namespace

// ------------------------------------------------------------------------------------------------
int GoDevice::init(int device, int games, int board_n, float komi, int action_size, int slots, int max_depth, hipStream_t stream, const int* const inv[8],
                   const int* const fwd[8], const uint64_t* keys)
{
    if (board_n < 2 || board_n > kGoMaxN || games < 1 || games > kRotPackGames || action_size != board_n * board_n + 1) {
        setError("GoDevice: unsupported shape (board %d, %d games, %d actions)", board_n, games, action_size);
        return MZ_ERR_ARG;
    }
    device_ = device;
    stream_ = stream;
    max_depth_ = max_depth;
    MZ_HIP(hipSetDevice(device));
    GoDevView& v = v_;
    v.games = games; v.n = board_n; v.P = board_n * board_n; v.W = (v.P + 63) / 64; v.A = action_size; v.slots = slots;
    v.Ppad = 64 * v.W; v.W32 = (v.P + 31) / 32; v.LW = (v.A + 63) / 64; v.komi = komi;
    const size_t GS = size_t(games) * slots;
    if (!h_snap_.alloc(games) || !d_snap_.alloc(games) || !stones_.alloc(GS * 2 * v.W) || !hash_.alloc(GS) || !meta_.alloc(GS * 2) ||
        !lab_.alloc(GS * v.Ppad) || !key_.alloc(size_t(2) * v.P) || !inv_.alloc(size_t(8) * v.P) || !fwd_.alloc(size_t(8) * v.A) ||
        !feat_.alloc(size_t(games) * 18 * v.W32) || !legal_.alloc(size_t(games) * v.LW) || !misc_i_.alloc(size_t(games) * 2) || !eval_.alloc(games)) {
        setError("GoDevice: allocation failed");
        return MZ_ERR_DEVICE;
    }
    memset(h_snap_.p, 0, games * sizeof(GoRootSnapshot));
    MZ_HIP(hipMemset(stones_.p, 0, stones_.n * sizeof(uint64_t)));
    MZ_HIP(hipMemset(lab_.p, 0, lab_.n * sizeof(uint16_t)));
    std::vector<uint16_t> t(size_t(8) * v.A);
    for (int r = 0; r < 8; ++r) { for (int p = 0; p < v.P; ++p) { t[size_t(r) * v.P + p] = static_cast<uint16_t>(inv[r][p]); } }
    MZ_HIP(hipMemcpy(inv_.p, t.data(), size_t(8) * v.P * sizeof(uint16_t), hipMemcpyHostToDevice));
    for (int r = 0; r < 8; ++r) { for (int a = 0; a < v.A; ++a) { t[size_t(r) * v.A + a] = static_cast<uint16_t>(fwd[r][a]); } }
    MZ_HIP(hipMemcpy(fwd_.p, t.data(), size_t(8) * v.A * sizeof(uint16_t), hipMemcpyHostToDevice));
    MZ_HIP(hipMemcpy(key_.p, keys, size_t(2) * v.P * sizeof(uint64_t), hipMemcpyHostToDevice));
    v.stones = stones_.p; v.hash = hash_.p; v.meta = meta_.p; v.lab = lab_.p; v.snap = d_snap_.p; v.key = key_.p; v.inv = inv_.p; v.fwd = fwd_.p;
    v.feat = feat_.p; v.legal = legal_.p; v.leaf_player = misc_i_.p; v.terminal = misc_i_.p + games; v.eval = eval_.p;
    return MZ_OK;
}

int GoDevice::uploadRoots()
{
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipMemcpyAsync(d_snap_.p, h_snap_.p, v_.games * sizeof(GoRootSnapshot), hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(go_root_kernel, dim3(v_.games), dim3(64), 0, stream_, v_);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int GoDevice::leafAsync(const PoolView& pv, const RotPack& rot, int slot)
{
    if (slot < 0 || slot >= v_.slots) { setError("GoDevice::leafAsync: slot %d out of range", slot); return MZ_ERR_ARG; }
    const size_t smem = sizeof(uint64_t) * (size_t(v_.Ppad) + pv.max_depth + 18 * v_.W) + size_t(v_.Ppad) * (4 + 2 + 1);
#define MZ_GO_CASE(K) \
    case K: hipLaunchKernelGGL(go_leaf_kernel<K>, dim3(v_.games), dim3(64), smem, stream_, v_, pv, rot, slot); break;
    switch (v_.W) {
        MZ_GO_CASE(1) MZ_GO_CASE(2) MZ_GO_CASE(3) MZ_GO_CASE(4) MZ_GO_CASE(5) MZ_GO_CASE(6)
    default: setError("GoDevice: board too large"); return MZ_ERR_ARG;
    }
#undef MZ_GO_CASE
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int GoDevice::candAsync(Pool& pool, const float* d_policy, const float* d_logit, const float* d_value, const RotPack& rot)
{
    const size_t smem = 2 * size_t(v_.A) * sizeof(Cand) + kSortStackBytes + 16;
    hipLaunchKernelGGL(az_cand_kernel, dim3(v_.games), dim3(64), smem, stream_, v_, d_policy, d_logit, d_value, rot, pool.d_cand_count_.p,
                       pool.d_cand_action_.p, pool.d_cand_policy_.p, pool.d_cand_logit_.p, pool.d_cand_player_.p, pool.d_value_.p, pool.d_reward_.p,
                       pool.errFlag());
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int GoDevice::readLeaf(uint32_t* feat, uint8_t* legal, int* terminal, float* eval, int* player)
{
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipStreamSynchronize(stream_));
    const int G = v_.games;
    if (feat) { MZ_HIP(hipMemcpy(feat, v_.feat, size_t(G) * 18 * v_.W32 * sizeof(uint32_t), hipMemcpyDeviceToHost)); }
    if (legal) {
        std::vector<uint64_t> w(size_t(G) * v_.LW);
        MZ_HIP(hipMemcpy(w.data(), v_.legal, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (int g = 0; g < G; ++g) {
            for (int a = 0; a < v_.A; ++a) { legal[size_t(g) * v_.A + a] = (w[size_t(g) * v_.LW + (a >> 6)] >> (a & 63)) & 1; }
        }
    }
    if (terminal) { MZ_HIP(hipMemcpy(terminal, v_.terminal, G * sizeof(int), hipMemcpyDeviceToHost)); }
    if (player) { MZ_HIP(hipMemcpy(player, v_.leaf_player, G * sizeof(int), hipMemcpyDeviceToHost)); }
    if (eval) { MZ_HIP(hipMemcpy(eval, v_.eval, G * sizeof(float), hipMemcpyDeviceToHost)); }
    return MZ_OK;
}

int sortCandidatesOnDevice(int device, const float* policy, int n, int* out_order)
{
    if (n < 1 || n > 4096) { setError("sortCandidatesOnDevice: n out of range"); return MZ_ERR_ARG; }
    MZ_HIP(hipSetDevice(device));
    DevBuf<float> dp;
    DevBuf<int> dord;
    if (!dp.alloc(n) || !dord.alloc(n + 1)) { setError("sortCandidatesOnDevice: allocation failed"); return MZ_ERR_DEVICE; }
    MZ_HIP(hipMemcpy(dp.p, policy, n * sizeof(float), hipMemcpyHostToDevice));
    MZ_HIP(hipMemset(dord.p + n, 0, sizeof(int)));
    hipLaunchKernelGGL(sort_test_kernel, dim3(1), dim3(64), 2 * size_t(n) * sizeof(Cand) + kSortStackBytes + 16, nullptr, dp.p, n, dord.p, dord.p + n);
    MZ_HIP(hipGetLastError());
    MZ_HIP(hipDeviceSynchronize());
    int err = 0;
    MZ_HIP(hipMemcpy(out_order, dord.p, n * sizeof(int), hipMemcpyDeviceToHost));
    MZ_HIP(hipMemcpy(&err, dord.p + n, sizeof(int), hipMemcpyDeviceToHost));
    if (err) { setError("sortCandidatesOnDevice: device sort failed"); return err; }
    return MZ_OK;
}

} // namespace mz
