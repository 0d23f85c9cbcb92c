// Device bodies of the device-resident Go leaf environment (see go_dev.h), shared by the stand-alone kernels (go_dev.hip) and the
// per-game simulation kernel (sim.hip).  Each body is run by ONE wave64 for game `g`; waveSync() orders its LDS traffic.
#pragma once
#include "go_dev.h"
#include "sort_emul.h"

#ifndef MZ_LPROF
#define MZ_LPROF(k) // experiment hook (sim.hip -DMZ_SIM_LPROF): time stamps inside the single-wave tree phases
#endif
#ifndef MZ_BPROF
#define MZ_BPROF(role, k) // experiment hook (sim.hip -DMZ_SIM_BPROF): time stamps inside the part of the leaf that runs beside the heads
#endif

namespace mz {

// single-wave phases: make the wave's LDS / global writes visible to its other lanes (no s_barrier: the other waves of a
// 512-thread workgroup are parked at a real barrier while wave 0 runs the tree phases of the simulation kernel)
__device__ __forceinline__ void waveSync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// (+ behind the arrays of the body: two words handed from PART 1 to PART 2, and a copy of the 2 x P Zobrist keys for the kernels that keep the block for a whole launch)
__host__ __device__ inline size_t goLeafKeyWord(int Ppad, int W, int max_depth) { return (sizeof(uint64_t) * (size_t(Ppad) + max_depth + 4 + 18 * W) + size_t(Ppad) * (4 + 2 + 1)) / sizeof(uint64_t) + 2; }
inline size_t goLeafSmemBytes(const GoDevView& v, int max_depth) { return sizeof(uint64_t) * (goLeafKeyWord(v.Ppad, v.W, max_depth) + 2 * size_t(v.P)); }

struct Cand { int action; float policy, logit; };
struct CandGreater { __host__ __device__ bool operator()(const Cand& l, const Cand& r) const { return l.policy > r.policy; } };
using CandSort = StdSortEmul<Cand, CandGreater>;
constexpr size_t kSortStackBytes = 3 * CandSort::kStack * sizeof(int);
inline size_t azCandSmemBytes(int A) { return 2 * size_t(A) * sizeof(Cand) + kSortStackBytes + 16; }

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

// position + legal mask + feature planes of the leaf selected for game `g` (one wave64; `smem` as sized by goLeafSmemBytes)
// PART 0: everything.  PART 1 / 2 (the one-game-per-CU simulation kernel, whose `smem` block then outlives the tower): 1 = what the network needs — the
// leaf's position (parent + move, stored to its slot), the history block of the planes, the player to move and the terminal flag; 2 = what only the
// phases AFTER the network need — the path's hashes, the groups' liberties and key sums, the legal mask, the score of a terminal leaf — run by another
// wave beside the heads: with SYNC it passes a workgroup barrier after the liberties and after the legal mask (the two barriers inside headsBody), and
// TWO waves share the work — ROLE 0: liberties + key sums | barrier | legal mask of the even 64-point chunks | barrier | score; ROLE 1: the path's hashes |
// barrier | legal mask of the odd chunks | barrier — so that each piece fits the interval of the heads it runs beside
template <int CPL, bool EXT_PLANES = false, int PART = 0, bool SYNC = false, int ROLE = 0>
// EXT_PLANES: the caller builds the feature planes itself from the history block this body leaves in `smem` (goPlanesPart, all waves)
// seen_lds: optional LDS copy of the root's positional-superko table (GoRootSnapshot::seen; constant during a move) — the simulation kernel
// makes one per launch so that the probes of every candidate point are LDS reads instead of dependent trips to L2
__device__ __forceinline__ void goLeafBody(const GoDevView& v, const PoolView& pv, int rot, int slot, int g, int lane, uint64_t* __restrict__ smem,
                                           const uint64_t* __restrict__ seen_lds = nullptr)
{
    const int P = v.P, n = v.n, W = v.W, Ppad = v.Ppad, MD = pv.max_depth;
    uint64_t* gh = smem;                                   // [Ppad] XOR of the keys of a group, by group id
    uint64_t* ph = gh + Ppad;                              // [MD + 4] (normalised) hashes of the positions along the path, 0-padded to a multiple of 4
    uint64_t* hb = ph + MD + 4;                            // [8][2][W] stones k moves before the leaf
    uint64_t* cur = hb + 16 * W;                           // [2][W] stones at the leaf
    int* libs = reinterpret_cast<int*>(cur + 2 * W);       // [Ppad] liberties of a group, by group id
    uint16_t* lab = reinterpret_cast<uint16_t*>(libs + Ppad); // [Ppad] group id per point
    uint8_t* col = reinterpret_cast<uint8_t*>(lab + Ppad);  // [Ppad] 0 empty, 1 black, 2 white, 3 off board
    uint64_t* stash = reinterpret_cast<uint64_t*>(col + Ppad); // [2] PART 1 -> PART 2: the leaf's hash; terminal flag, player to move, move / pass counters (Ppad is a multiple of 64: aligned)
    // the Zobrist keys: with PART 1 / 2 the caller keeps a copy behind the stash (LDS) — a key is on the walk's critical path at every move
    const uint64_t* zkey = PART != 0 ? smem + goLeafKeyWord(Ppad, W, MD) : v.key;

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
    uint64_t hash = 0;
    int nmoves = 0, passes = 0;
    int t = (depth & 1) ? 3 - root_turn : root_turn; // the player to move at the leaf
    bool terminal = false;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int p = i * 64 + lane;
        nb[i][0] = nb[i][1] = nb[i][2] = nb[i][3] = -1;
        if (p < P) {
            const int x = p % n, y = p / n;
            if (y + 1 < n) { nb[i][0] = static_cast<short>(p + n); }
            if (x + 1 < n) { nb[i][1] = static_cast<short>(p + 1); }
            if (y > 0) { nb[i][2] = static_cast<short>(p - n); }
            if (x > 0) { nb[i][3] = static_cast<short>(p - 1); }
        }
    }
    if constexpr (PART == 2) { // the leaf as PART 1 left it in the block
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = i * 64 + lane;
            c[i] = col[p];
            l[i] = lab[p];
            sbw[i] = cur[i];
            sww[i] = cur[W + i];
        }
        hash = stash[0];
        terminal = (stash[1] & 1) != 0;
        t = static_cast<int>((stash[1] >> 1) & 3); // (no trip to the root's snapshot in global memory)
        passes = static_cast<int>((stash[1] >> 4) & 15);
        nmoves = static_cast<int>(stash[1] >> 8);
    }
    if constexpr (PART != 2) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int p = i * 64 + lane;
        sbw[i] = v.stones[((sb + src) * 2 + 0) * W + i];
        sww[i] = v.stones[((sb + src) * 2 + 1) * W + i];
        c[i] = 3;
        l[i] = 0;
        if (p < P) {
            c[i] = ((sbw[i] >> lane) & 1) ? 1 : (((sww[i] >> lane) & 1) ? 2 : 0);
            l[i] = v.lab[(sb + src) * Ppad + p];
        }
        col[p] = static_cast<uint8_t>(c[i]);
        lab[p] = static_cast<uint16_t>(l[i]);
    }
    hash = v.hash[sb + src];
    nmoves = v.meta[(sb + src) * 2]; passes = v.meta[(sb + src) * 2 + 1];
    waveSync();
    MZ_LPROF(1); // parent slot loaded

    if (depth >= 1) { // leaf = parent + one move (ref go.cpp:132-190, observable effects only)
        const int a = pact[len - 1], m = 3 - t;
        ++nmoves;
        hash ^= v.turn_key; // situational superko: every move, pass included (ref go.cpp:141); 0 with the positional rule
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
            waveSync();
            // place the stone; the own groups it touches become one group whose id is the new point (unused as an id: it was empty)
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int p = i * 64 + lane;
                if (p == a) { c[i] = m; l[i] = a; }
                else if (c[i] == m && (l[i] == own[0] || l[i] == own[1] || l[i] == own[2] || l[i] == own[3])) { l[i] = a; }
                col[p] = static_cast<uint8_t>(c[i]);
                lab[p] = static_cast<uint16_t>(l[i]);
            }
            waveSync();
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
                        hx ^= zkey[size_t(2 - m) * P + p];
                    }
                }
                hx = waveXor64(hx);
            }
            hash ^= zkey[size_t(m - 1) * P + a] ^ hx;
            waveSync();
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = i * 64 + lane;
            sbw[i] = __ballot(c[i] == 1);
            sww[i] = __ballot(c[i] == 2);
            if constexpr (PART != 1) { // (PART 1 leaves the slot's store to PART 2: the next fence would wait for it)
                if (lane == 0) {
                    v.stones[((sb + slot) * 2 + 0) * W + i] = sbw[i];
                    v.stones[((sb + slot) * 2 + 1) * W + i] = sww[i];
                }
                if (p < P) { v.lab[(sb + slot) * Ppad + p] = static_cast<uint16_t>(l[i]); }
            }
        }
        if constexpr (PART != 1) {
            if (lane == 0) {
                v.hash[sb + slot] = hash;
                v.meta[(sb + slot) * 2] = nmoves;
                v.meta[(sb + slot) * 2 + 1] = passes;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        if (lane == 0) { cur[i] = sbw[i]; cur[W + i] = sww[i]; }
    }
    MZ_LPROF(2); // move applied, slot stored
    terminal = passes >= 2 || nmoves > 2 * P; // ref go.cpp:246-257
    if (lane == 0) { stash[0] = hash; stash[1] = (terminal ? 1 : 0) | (static_cast<uint64_t>(t) << 1) | (static_cast<uint64_t>(passes) << 4) | (static_cast<uint64_t>(nmoves) << 8); }
    if constexpr (PART == 1) { // (PART 2 counts the liberties on two waves: the counters are cleared here)
#pragma unroll
        for (int i = 0; i < CPL; ++i) { libs[i * 64 + lane] = 0; gh[i * 64 + lane] = 0; }
    }
    waveSync();
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
    waveSync();
    if constexpr (!EXT_PLANES) {
        const uint16_t* map = v.inv + size_t(rot) * P;
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
    MZ_LPROF(5); // planes
    if (lane == 0) {
        v.leaf_player[g] = t;
        v.terminal[g] = terminal ? 1 : 0;
    }
    } // PART != 2
    if constexpr (PART == 1) { return; }
    constexpr bool kShared = PART == 2 && SYNC; // two waves share PART 2 (ROLE)
    static_assert(PART != 2 || SYNC, "PART 2 is the two-wave version (it also stores the leaf's slot)");
    MZ_BPROF(ROLE, 0);
    // the legal mask of the 64-point chunks i with want(i) (ref go.cpp:208-244): not occupied, not suicide, not a positional-superko repeat
    auto legalMask = [&](auto want) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            if (!want(i)) { continue; }
            bool bit = false;
            if (c[i] == 0 && !terminal) {
                const int p = i * 64 + lane;
                bool ok = false;
                uint64_t nh = hash ^ v.turn_key ^ zkey[size_t(t - 1) * P + p];
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
                    const uint64_t* seen = seen_lds ? seen_lds : S.seen;
                    for (uint32_t s = static_cast<uint32_t>(h) & (kGoSeenCap - 1);; s = (s + 1) & (kGoSeenCap - 1)) {
                        const uint64_t e = seen[s];
                        if (e == 0) { break; }
                        if (e == h) { rep = true; break; }
                    }
                    for (int d = 0; d < depth; d += 4) { // four hashes per step, no early exit: a deep path made this loop the longest part of the leaf
                        const uint64_t a0 = ph[d], a1 = ph[d + 1], a2 = ph[d + 2], a3 = ph[d + 3];
                        rep |= (a0 == h) | (a1 == h) | (a2 == h) | (a3 == h);
                    }
                    bit = !rep;
                }
            }
            uint64_t w = __ballot(bit);
            if (i == (P >> 6)) { w |= 1ull << (P & 63); } // pass is always legal
            if (lane == 0) { v.legal[size_t(g) * v.LW + i] = w; }
        }
    };
    // Shared by two waves, the pieces are laid into the three intervals of the heads (conv1x1 | FCs | softmax + value: 2.2 / 2.9 / 2.2 us on BASELINE configs[1]):
    //   ROLE 0: liberties of the even chunks              | barrier | key sums, legal mask of the even chunks | barrier | score of a terminal leaf, the leaf's scalars
    //   ROLE 1: path hashes, liberties of the odd chunks  | barrier |                                        | barrier | legal mask of the odd chunks
    // ---- hashes along the path (d = 1 .. depth; the root and everything before it is in the root's table) ----
    if (!kShared || ROLE == 1) {
        for (int d = 1 + lane; d <= depth; d += 64) { ph[d - 1] = normH(d == depth ? hash : v.hash[sb + hs[path[d]]]); }
        if (lane < 4) { ph[depth + lane] = 0; } // normH never returns 0: the padding matches no candidate
    }
    // ---- group liberties / key sums at the leaf ----
    if constexpr (!kShared) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) { libs[i * 64 + lane] = 0; gh[i * 64 + lane] = 0; }
        waveSync();
    }
    {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            if (kShared && (i & 1) != ROLE) { continue; }
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
    }
    waveSync();
    MZ_BPROF(ROLE, 1);
    if constexpr (SYNC) { __syncthreads(); }
    MZ_BPROF(ROLE, 2);
    if constexpr (kShared && ROLE == 1) { // the leaf's position into its slab slot (PART 1 left it out)
        if (depth >= 1) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int p = i * 64 + lane;
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
    }
    if (!kShared || ROLE == 0) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = i * 64 + lane;
            if (c[i] == 3 - t && libs[l[i]] == 1) { atomicXor(reinterpret_cast<unsigned long long*>(&gh[l[i]]), static_cast<unsigned long long>(zkey[size_t(2 - t) * P + p])); }
        }
        waveSync();
        MZ_LPROF(3); // path hashes, liberties, key sums
        if constexpr (kShared) { legalMask([](int i) { return (i & 1) == 0; }); }
        else { legalMask([](int) { return true; }); }
        if (v.LW > CPL && lane == 0) { v.legal[size_t(g) * v.LW + CPL] = (P >> 6) == CPL ? 1ull << (P & 63) : 0; } // P a multiple of 64
        MZ_LPROF(4); // legal mask
    }
    MZ_BPROF(ROLE, 3);
    if constexpr (SYNC) { __syncthreads(); }
    MZ_BPROF(ROLE, 4);
    if constexpr (kShared && ROLE == 1) {
        legalMask([](int i) { return (i & 1) == 1; });
        return;
    }
    // ---- terminal: Tromp-Taylor area score + komi (ref go.cpp:259-278,703-723) ----
    float eval = 0.0f;
    if (terminal) {
        waveSync();
        uint16_t* rl = lab; // region id of the empty points: the smallest point of the region
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = i * 64 + lane;
            rl[p] = static_cast<uint16_t>(p);
            libs[p] = 0;
            gh[p] = 0;
        }
        waveSync();
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
            waveSync();
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
        waveSync();
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

// The planes of goLeafBody<CPL, true>, shared by `nw` waves (the per-game simulation kernel: every wave takes the planes ch = w, w + nw, ...):
// planes 2k / 2k+1 = own / opponent stones k moves ago under the rotation, 16 / 17 = black / white to move (ref go.cpp:280-308).
// `smem` is the leaf's scratch block (its history words were filled by the leaf body), the player to move is read from v.leaf_player.
template <int CPL>
__device__ __forceinline__ void goPlanesPart(const GoDevView& v, int max_depth, int rot, int g, int w, int nw, int lane, const uint64_t* __restrict__ smem)
{
    const int P = v.P, W = v.W;
    const uint64_t* hb = smem + v.Ppad + max_depth + 4;
    const int t = v.leaf_player[g];
    const uint16_t* map = v.inv + size_t(rot) * P;
    uint32_t* out = v.feat + size_t(g) * 18 * v.W32;
    for (int ch = w; ch < 18; ch += nw) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = i * 64 + lane;
            uint64_t word;
            if (ch < 16) {
                const int q = p < P ? map[p] : 0;
                const int k = ch >> 1, color = (ch & 1) == 0 ? t - 1 : 2 - t;
                word = __ballot(p < P && ((hb[(k * 2 + color) * W + (q >> 6)] >> (q & 63)) & 1));
            } else {
                word = (ch == 16 ? t == 1 : t == 2) ? __ballot(p < P) : 0;
            }
            if (lane == 0) {
                if (2 * i < v.W32) { out[ch * v.W32 + 2 * i] = static_cast<uint32_t>(word); }
                if (2 * i + 1 < v.W32) { out[ch * v.W32 + 2 * i + 1] = static_cast<uint32_t>(word >> 32); }
            }
        }
    }
}

// ---- Othello (ref environment/othello/othello.cpp:61-140,195-255): the whole position is two 64-bit boards, so every lane carries
// it and the rules are scalar bit arithmetic; only the rotated feature planes are built with ballots.  Slot layout: the `stones`
// words of the Go slab hold the two boards, `meta` (moves played, trailing passes).
struct OthMasks { unsigned long long full, not_left, not_right; int n; };
__device__ __forceinline__ unsigned long long othShift(const OthMasks& k, unsigned long long b, int d)
{
    switch (d) {
    case 0: return (b << k.n) & k.full;
    case 1: return b >> k.n;
    case 2: return (b & k.not_left) >> 1;
    case 3: return ((b & k.not_right) << 1) & k.full;
    case 4: return ((b & k.not_left) << (k.n - 1)) & k.full;
    case 5: return ((b & k.not_right) << (k.n + 1)) & k.full;
    case 6: return (b & k.not_right) >> (k.n - 1);
    default: return (b & k.not_left) >> (k.n + 1);
    }
}
__device__ __forceinline__ unsigned long long othMoves(const OthMasks& k, unsigned long long me, unsigned long long op)
{
    const unsigned long long empty = k.full & ~(me | op);
    unsigned long long m = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        unsigned long long x = othShift(k, me, d) & op;
        for (int i = 0; i < k.n - 3; ++i) { x |= othShift(k, x, d) & op; }
        m |= othShift(k, x, d) & empty;
    }
    return m;
}

__device__ __forceinline__ void othLeafBody(const GoDevView& v, const PoolView& pv, int rot, int slot, int g, int lane)
{
    const int P = v.P, n = v.n, MD = pv.max_depth;
    OthMasks k;
    k.n = n;
    k.full = P == 64 ? ~0ull : ((1ull << P) - 1);
    k.not_left = 0; k.not_right = 0;
    for (int y = 0; y < n; ++y) {
        const unsigned long long row = ((1ull << n) - 1) << (y * n);
        k.not_left |= row & ~(1ull << (y * n));
        k.not_right |= row & ~(1ull << (y * n + n - 1));
    }
    const int len = pv.path_len[g];
    const int* path = pv.path + size_t(g) * MD;
    const int* pact = pv.path_action + size_t(g) * MD;
    const int depth = len - 1;
    const GoRootSnapshot& S = v.snap[g];
    const int root_turn = S.turn;
    const size_t sb = size_t(g) * v.slots;
    const int* hs = pv.hslot + size_t(g) * pv.cap;
    const int src = depth == 0 ? 0 : hs[path[len - 2]];
    unsigned long long s[2] = {v.stones[((sb + src) * 2 + 0) * v.W], v.stones[((sb + src) * 2 + 1) * v.W]};
    int nmoves = v.meta[(sb + src) * 2], passes = v.meta[(sb + src) * 2 + 1];
    const int t = (depth & 1) ? 3 - root_turn : root_turn; // the player to move at the leaf
    if (depth >= 1) {
        const int a = pact[len - 1], m = 3 - t;
        ++nmoves;
        if (a >= P) {
            passes = passes + 1 > 2 ? 2 : passes + 1;
        } else {
            passes = 0;
            unsigned long long& me = s[m - 1];
            unsigned long long& op = s[2 - m];
            const unsigned long long placed = 1ull << a;
            unsigned long long flip = 0;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                unsigned long long line = 0, x = othShift(k, placed, d);
                for (int i = 0; i < n && (x & op); ++i) { line |= x; x = othShift(k, x, d); }
                if (x & me) { flip |= line; }
            }
            me |= placed | flip;
            op &= ~flip;
        }
        if (lane == 0) {
            v.stones[((sb + slot) * 2 + 0) * v.W] = s[0];
            v.stones[((sb + slot) * 2 + 1) * v.W] = s[1];
            v.meta[(sb + slot) * 2] = nmoves;
            v.meta[(sb + slot) * 2 + 1] = passes;
        }
    }
    const bool terminal = passes >= 2; // ref othello.cpp: two consecutive passes
    const unsigned long long mv = terminal ? 0ull : othMoves(k, s[t - 1], s[2 - t]);
    if (lane == 0) { // legal mask: the moves, or pass when there is none (ref othello.cpp:195-201)
        unsigned long long w0 = mv, w1 = 0;
        if (!terminal && mv == 0) { if (P < 64) { w0 |= 1ull << P; } else { w1 = 1ull; } }
        v.legal[size_t(g) * v.LW] = w0;
        if (v.LW > 1) { v.legal[size_t(g) * v.LW + 1] = w1; }
    }
    { // planes (ref othello.cpp:238-262): own, opponent, black to move, white to move — under the cycle's rotation
        const uint16_t* map = v.inv + size_t(rot) * P;
        const int q = lane < P ? map[lane] : 0;
        const unsigned long long own = __ballot(lane < P && ((s[t - 1] >> q) & 1)), opp = __ballot(lane < P && ((s[2 - t] >> q) & 1));
        const unsigned long long ones = __ballot(lane < P);
        unsigned long long mine = 0;
        if (lane == 0) { mine = own; }
        if (lane == 1) { mine = opp; }
        if (lane == 2) { mine = t == 1 ? ones : 0; }
        if (lane == 3) { mine = t == 2 ? ones : 0; }
        uint32_t* out = v.feat + size_t(g) * 4 * v.W32;
        if (lane < 4) {
            out[lane * v.W32] = static_cast<uint32_t>(mine);
            if (v.W32 > 1) { out[lane * v.W32 + 1] = static_cast<uint32_t>(mine >> 32); }
        }
    }
    float eval = 0.0f;
    if (terminal) { // ref othello.cpp:211-236: a winner only once neither side can move
        if (othMoves(k, s[0], s[1]) == 0 && othMoves(k, s[1], s[0]) == 0) {
            const int b = __popcll(s[0]), w = __popcll(s[1]);
            eval = b > w ? 1.0f : (b < w ? -1.0f : 0.0f);
        }
    }
    if (lane == 0) {
        v.leaf_player[g] = t;
        v.terminal[g] = terminal ? 1 : 0;
        v.eval[g] = eval;
    }
}

// ---- TicTacToe (ref environment/tictactoe/tictactoe.cpp:19-97): two 9-bit boards per slot; the planes are Othello's (own, opponent,
// black to move, white to move); no pass, terminal = a complete line or a full board.
__device__ __forceinline__ void tttLeafBody(const GoDevView& v, const PoolView& pv, int rot, int slot, int g, int lane)
{
    const int P = 9, MD = pv.max_depth;
    const int len = pv.path_len[g];
    const int* path = pv.path + size_t(g) * MD;
    const int* pact = pv.path_action + size_t(g) * MD;
    const int depth = len - 1;
    const GoRootSnapshot& S = v.snap[g];
    const int root_turn = S.turn;
    const size_t sb = size_t(g) * v.slots;
    const int* hs = pv.hslot + size_t(g) * pv.cap;
    const int src = depth == 0 ? 0 : hs[path[len - 2]];
    unsigned s[2] = {static_cast<unsigned>(v.stones[((sb + src) * 2 + 0) * v.W]), static_cast<unsigned>(v.stones[((sb + src) * 2 + 1) * v.W])};
    const int t = (depth & 1) ? 3 - root_turn : root_turn; // the player to move at the leaf
    if (depth >= 1) {
        s[2 - t] |= 1u << pact[len - 1]; // moved by the other player
        if (lane == 0) {
            v.stones[((sb + slot) * 2 + 0) * v.W] = s[0];
            v.stones[((sb + slot) * 2 + 1) * v.W] = s[1];
        }
    }
    int winner = 0;
    {
        const unsigned lines[8] = {0007, 0070, 0700, 0111, 0222, 0444, 0421, 0124};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (winner == 0 && (s[0] & lines[i]) == lines[i]) { winner = 1; }
            if (winner == 0 && (s[1] & lines[i]) == lines[i]) { winner = 2; }
        }
    }
    const bool terminal = winner != 0 || (s[0] | s[1]) == 0777u;
    if (lane == 0) { v.legal[size_t(g) * v.LW] = terminal ? 0ull : static_cast<unsigned long long>(~(s[0] | s[1]) & 0777u); }
    {
        const uint16_t* map = v.inv + size_t(rot) * P;
        const int q = lane < P ? map[lane] : 0;
        const unsigned own = static_cast<unsigned>(__ballot(lane < P && ((s[t - 1] >> q) & 1)));
        const unsigned opp = static_cast<unsigned>(__ballot(lane < P && ((s[2 - t] >> q) & 1)));
        uint32_t* out = v.feat + size_t(g) * 4 * v.W32;
        if (lane == 0) { out[0] = own; }
        if (lane == 1) { out[v.W32] = opp; }
        if (lane == 2) { out[2 * v.W32] = t == 1 ? 0777u : 0u; }
        if (lane == 3) { out[3 * v.W32] = t == 2 ? 0777u : 0u; }
    }
    if (lane == 0) {
        v.leaf_player[g] = t;
        v.terminal[g] = terminal ? 1 : 0;
        v.eval[g] = winner == 1 ? 1.0f : (winner == 2 ? -1.0f : 0.0f);
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
    waveSync();
    if (k > 16 && __ballot(tie) != 0) { // ties among > 16 elements: only the exact introsort gives the reference's order
        if (lane == 0) {
            StdSortEmul<Cand, CandGreater> s{cs, CandGreater()};
            if (!s.sort(k, stack) && err) { atomicExch(err, MZ_ERR_CAPACITY); }
        }
        waveSync();
        for (int i = lane; i < k; i += 64) { out[i] = cs[i]; }
        waveSync();
    }
}

// The same order computed by ALL waves of a workgroup (the per-game simulation kernel: 7 of its 8 waves idle while wave 0 runs the tree
// phases, and the rank sort is VALU-bound: k^2 comparisons — 3.7 of the 8.5 us of candidates + expand + backup at 82 candidates).
//   candDense   (one wave)   policies of cs[0..k) into a dense float array `dense` [128], padding below every softmax output
//   candRankPart(every wave) wave w of nw counts, for each candidate i, the candidates j of ITS share of j that rank before i / tie with i
//   candScatter (one wave)   sums the partial counts, writes out[rank] = cs[i]; ties among > 16 candidates -> the exact introsort replay
// Workgroup barriers between the three steps are the caller's.  k <= 128.
constexpr int kCandCoopMax = 128;
__device__ __forceinline__ void candDense(const Cand* cs, int k, int lane, float* dense)
{
    dense[lane] = lane < k ? cs[lane].policy : -3.402823466e+38f;
    dense[64 + lane] = 64 + lane < k ? cs[64 + lane].policy : -3.402823466e+38f;
}
__device__ __forceinline__ void candRankPart(const float* dense, int k, int w, int nw, int lane, int* part /* [nw][4][64] */)
{
    const float p0 = dense[lane], p1 = dense[64 + lane];
    const int groups = (k + 3) >> 2, per = (32 + nw - 1) / nw;
    int r0 = 0, r1 = 0, e0 = 0, e1 = 0;
    for (int q = w * per; q < (w + 1) * per && q < groups; ++q) {
        const float4 d = reinterpret_cast<const float4*>(dense)[q];
        const float dj[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * q + e;
            r0 += (dj[e] > p0) || (dj[e] == p0 && j < lane);
            r1 += (dj[e] > p1) || (dj[e] == p1 && j < 64 + lane);
            e0 += (dj[e] == p0);
            e1 += (dj[e] == p1);
        }
    }
    int* mine = part + w * 256;
    mine[lane] = r0; mine[64 + lane] = r1; mine[128 + lane] = e0; mine[192 + lane] = e1;
}
__device__ __forceinline__ void candScatter(Cand* cs, Cand* out, int* stack, int k, int nw, int lane, const int* part, int* err)
{
    int r0 = 0, r1 = 0, e0 = 0, e1 = 0;
    for (int w = 0; w < nw; ++w) { const int* p = part + w * 256; r0 += p[lane]; r1 += p[64 + lane]; e0 += p[128 + lane]; e1 += p[192 + lane]; }
    const bool has0 = lane < k, has1 = 64 + lane < k;
    const bool tie = (has0 && e0 > 1) || (has1 && e1 > 1); // every candidate equals itself once
    if (has0) { out[r0] = cs[lane]; }
    if (has1) { out[r1] = cs[64 + lane]; }
    waveSync();
    if (k > 16 && __ballot(tie) != 0) { // ties among > 16 elements: only the exact introsort gives the reference's order
        if (lane == 0) {
            StdSortEmul<Cand, CandGreater> s{cs, CandGreater()};
            if (!s.sort(k, stack) && err) { atomicExch(err, MZ_ERR_CAPACITY); }
        }
        waveSync();
        for (int i = lane; i < k; i += 64) { out[i] = cs[i]; }
        waveSync();
    }
}

// ... and for boards of more than 128 actions (13x13 / 19x19 Go: up to 362 candidates, whose rank sort on ONE wave is k^2 / 64 = 2 000 dependent LDS reads
// per lane): the same three steps with up to six candidates per lane.  dense [384] floats, part [nw][2][384] counts.  k <= 384.
constexpr int kCandCoopMaxW = 384;
__device__ __forceinline__ void candDenseW(const Cand* cs, int k, int lane, float* dense)
{
#pragma unroll
    for (int c = 0; c < kCandCoopMaxW / 64; ++c) { const int i = 64 * c + lane; dense[i] = i < k ? cs[i].policy : -3.402823466e+38f; }
}
__device__ __forceinline__ void candRankPartW(const float* dense, int k, int w, int nw, int lane, int* part /* [nw][2][384] */)
{
    constexpr int NC = kCandCoopMaxW / 64;
    float p[NC];
    int r[NC], eq[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { p[c] = dense[64 * c + lane]; r[c] = 0; eq[c] = 0; }
    const int groups = (k + 3) >> 2, per = (groups + nw - 1) / nw;
    for (int q = w * per; q < (w + 1) * per && q < groups; ++q) {
        const float4 d = reinterpret_cast<const float4*>(dense)[q];
        const float dj[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * q + e;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                r[c] += (dj[e] > p[c]) || (dj[e] == p[c] && j < 64 * c + lane);
                eq[c] += (dj[e] == p[c]);
            }
        }
    }
    int* mine = part + w * 2 * kCandCoopMaxW;
#pragma unroll
    for (int c = 0; c < NC; ++c) { mine[64 * c + lane] = r[c]; mine[kCandCoopMaxW + 64 * c + lane] = eq[c]; }
}
__device__ __forceinline__ void candScatterW(Cand* cs, Cand* out, int* stack, int k, int nw, int lane, const int* part, int* err)
{
    constexpr int NC = kCandCoopMaxW / 64;
    bool tie = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = 64 * c + lane;
        int r = 0, e = 0;
        for (int w = 0; w < nw; ++w) { const int* p = part + w * 2 * kCandCoopMaxW; r += p[i]; e += p[kCandCoopMaxW + i]; }
        if (i < k) { out[r] = cs[i]; tie = tie || e > 1; } // every candidate equals itself once
    }
    waveSync();
    if (k > 16 && __ballot(tie) != 0) { // ties among > 16 elements: only the exact introsort gives the reference's order
        if (lane == 0) {
            StdSortEmul<Cand, CandGreater> s{cs, CandGreater()};
            if (!s.sort(k, stack) && err) { atomicExch(err, MZ_ERR_CAPACITY); }
        }
        waveSync();
        for (int i = lane; i < k; i += 64) { out[i] = cs[i]; }
        waveSync();
    }
}

// AlphaZero candidates of a leaf (ref zero_actor.cpp:215-245): legal actions in action order, policy / logit looked up through
// the rotation, sorted by policy; a terminal leaf has no children and its value is the game result (zero_actor.cpp:85)
// step 1: the legal actions of the leaf, in action order, into cs[]; returns their number (0 at a terminal leaf)
__device__ __forceinline__ int azCandGather(const GoDevView& v, const float* __restrict__ policy, const float* __restrict__ logit, int rot, int g, int lane,
                                            Cand* __restrict__ cs)
{
    const int A = v.A;
    int k = 0;
    if (v.terminal[g] != 0) { return 0; }
    const uint16_t* fwd = v.fwd + size_t(rot) * A;
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
    return k;
}
// step 3: the sorted candidates and the leaf's scalars where expand + backup read them
__device__ __forceinline__ void azCandStore(const GoDevView& v, const float* __restrict__ value, const Cand* __restrict__ out, int k, int* __restrict__ cand_count,
                                            int* __restrict__ cand_action, float* __restrict__ cand_policy, float* __restrict__ cand_logit,
                                            int* __restrict__ cand_player, float* __restrict__ value_out, float* __restrict__ reward_out, int g, int lane)
{
    const int A = v.A;
    for (int i = lane; i < k; i += 64) {
        cand_action[size_t(g) * A + i] = out[i].action;
        cand_policy[size_t(g) * A + i] = out[i].policy;
        cand_logit[size_t(g) * A + i] = out[i].logit;
    }
    if (lane == 0) {
        const bool terminal = v.terminal[g] != 0;
        cand_count[g] = k;
        cand_player[g] = v.leaf_player[g];
        value_out[g] = terminal ? v.eval[g] : value[g];
        reward_out[g] = 0.0f;
    }
}
__device__ __forceinline__ void azCandBody(const GoDevView& v, const float* __restrict__ policy, const float* __restrict__ logit,
                                           const float* __restrict__ value, int rot, int* __restrict__ cand_count, int* __restrict__ cand_action,
                                           float* __restrict__ cand_policy, float* __restrict__ cand_logit, int* __restrict__ cand_player,
                                           float* __restrict__ value_out, float* __restrict__ reward_out, int* __restrict__ err, int g, int lane,
                                           uint64_t* __restrict__ smem)
{
    Cand* cs = reinterpret_cast<Cand*>(smem);
    Cand* out = cs + v.A;
    int* stack = reinterpret_cast<int*>(out + v.A);
    const int k = azCandGather(v, policy, logit, rot, g, lane, cs);
    if (k > 0) {
        waveSync();
        orderCandidates(cs, out, stack, k, lane, err);
    }
    azCandStore(v, value, out, k, cand_count, cand_action, cand_policy, cand_logit, cand_player, value_out, reward_out, g, lane);
}

// scratch of the cooperative variant behind the cs / out / stack block: dense[128] floats + nw x 256 partial counts
inline size_t candCoopOffsetBytes(int A) { return (azCandSmemBytes(A) + 15) & ~size_t(15); }
inline size_t candCoopSmemBytes(int A, int nw) { return candCoopOffsetBytes(A) + kCandCoopMax * sizeof(float) + size_t(nw) * 256 * sizeof(int); }
inline size_t candCoopSmemBytesW(int A, int nw) { return candCoopOffsetBytes(A) + kCandCoopMaxW * sizeof(float) + size_t(nw) * 2 * kCandCoopMaxW * sizeof(int); }

} // namespace mz
