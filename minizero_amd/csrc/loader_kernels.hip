// Device side of the learner-side sampler (loader.cpp): the feature planes of a sampled (game, position) are produced ON the GPU, straight into
// the training batch in HBM.  The reference replays every sampled game from its first move on a fresh host environment
// (ref environment/base/base_env.h:235-241 BaseEnvLoader::getFeatures, called per sample by learner/data_loader.cpp:141-151,163-170).
// Here one wave64 per sample replays its game on the device rules engine the self-play worker already has (go_body.h: one position slot per
// move, leaf = parent slot + one move), for all samples of the batch at once; Atari-shaped samples expand their 8 stored screens.
#include "go_body.h"
#include "loader_dev.h"

namespace mz {

namespace {


// Go: the moves 1 .. upto of sample g applied ONE AFTER THE OTHER ON THE SAME WAVE STATE (stones and group ids in registers + LDS, the hash in a register)
// instead of one goLeafBody per move through the position slots in global memory (load the parent, apply, store: 3 us per move, 0.5 ms for a
// 160-move sample).  Same observable effects as the "leaf = parent + one move" part of goLeafBody (ref go.cpp:132-190); every position's hash and
// move counters are stored, stones and group ids only for the last `keep` positions — what the planes of the sampled position (8 positions of
// history) and the final goLeafBody (its parent slot) read.
template <int CPL>
__device__ __forceinline__ void goReplayMoves(const GoDevView& v, const PoolView& pv, int g, int lane, uint64_t* __restrict__ smem, int upto, int keep)
{
    const int P = v.P, n = v.n, W = v.W, Ppad = v.Ppad, MD = pv.max_depth;
    uint64_t* gh = smem;
    uint64_t* ph = gh + Ppad;
    uint64_t* hb = ph + MD + 4;
    uint64_t* cur = hb + 16 * W;
    int* libs = reinterpret_cast<int*>(cur + 2 * W);
    uint16_t* lab = reinterpret_cast<uint16_t*>(libs + Ppad); // same LDS layout as goLeafBody
    uint8_t* col = reinterpret_cast<uint8_t*>(lab + Ppad);
    const int* pact = pv.path_action + size_t(g) * MD;
    const size_t sb = size_t(g) * v.slots;
    const int root_turn = v.snap[g].turn;
    int c[CPL], l[CPL];
    short nb[CPL][4];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int p = i * 64 + lane;
        const uint64_t sbw = v.stones[((sb + 0) * 2 + 0) * W + i], sww = v.stones[((sb + 0) * 2 + 1) * W + i];
        c[i] = 3;
        l[i] = 0;
        nb[i][0] = nb[i][1] = nb[i][2] = nb[i][3] = -1;
        if (p < P) {
            c[i] = ((sbw >> lane) & 1) ? 1 : (((sww >> lane) & 1) ? 2 : 0);
            l[i] = v.lab[(sb + 0) * Ppad + p];
            const int x = p % n, y = p / n;
            if (y + 1 < n) { nb[i][0] = static_cast<short>(p + n); }
            if (x + 1 < n) { nb[i][1] = static_cast<short>(p + 1); }
            if (y > 0) { nb[i][2] = static_cast<short>(p - n); }
            if (x > 0) { nb[i][3] = static_cast<short>(p - 1); }
        }
        col[p] = static_cast<uint8_t>(c[i]);
        lab[p] = static_cast<uint16_t>(l[i]);
    }
    uint64_t hash = v.hash[sb + 0];
    int nmoves = v.meta[(sb + 0) * 2], passes = v.meta[(sb + 0) * 2 + 1];
    waveSync();
    for (int d = 1; d <= upto; ++d) {
        const int t = (d & 1) ? 3 - root_turn : root_turn; // the player to move AFTER move d
        const int a = pact[d], m = 3 - t;
        ++nmoves;
        hash ^= v.turn_key;
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
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int p = i * 64 + lane;
                if (p == a) { c[i] = m; l[i] = a; }
                else if (c[i] == m && (l[i] == own[0] || l[i] == own[1] || l[i] == own[2] || l[i] == own[3])) { l[i] = a; }
                col[p] = static_cast<uint8_t>(c[i]);
                lab[p] = static_cast<uint16_t>(l[i]);
            }
            waveSync();
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
            waveSync();
        }
        if (lane == 0) {
            v.hash[sb + d] = hash;
            v.meta[(sb + d) * 2] = nmoves;
            v.meta[(sb + d) * 2 + 1] = passes;
        }
        if (d > upto - keep) { // one of the last positions: the planes of the sampled position and the final goLeafBody read its slot
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int p = i * 64 + lane;
                const uint64_t sbw = __ballot(c[i] == 1), sww = __ballot(c[i] == 2);
                if (lane == 0) {
                    v.stones[((sb + d) * 2 + 0) * W + i] = sbw;
                    v.stones[((sb + d) * 2 + 1) * W + i] = sww;
                }
                if (p < P) { v.lab[(sb + d) * Ppad + p] = static_cast<uint16_t>(l[i]); }
            }
        }
    }
    waveSync();
}

// sample g: moves pact[g][1 .. pos[g]] replayed from the root snapshot (slot 0); the planes of the last position under rot[g] end in v.feat
template <int KIND, int CPL>
__global__ __launch_bounds__(64) void replay_kernel(GoDevView v, PoolView pv, const int* __restrict__ pos, const uint8_t* __restrict__ rot)
{
    extern __shared__ uint64_t smem[];
    const int g = blockIdx.x, lane = threadIdx.x;
    const int n = pos[g], r = rot[g];
    int d0 = 0;
    if constexpr (KIND == 0) { // Go: all moves but the last on one wave state; the last one (the sampled position) through goLeafBody
        if (n > 1) { goReplayMoves<CPL>(v, pv, g, lane, smem, n - 1, 8); d0 = n; }
    }
    for (int d = d0; d <= n; ++d) {
        if (lane == 0) { pv.path_len[g] = d + 1; }
        waveSync();
        if (d == 0 && n > 0) { continue; } // slot 0 is the uploaded root: only evaluate it when it is the sampled position itself
        if constexpr (KIND == 2) { tttLeafBody(v, pv, r, d, g, lane); }
        else if constexpr (KIND == 1) { othLeafBody(v, pv, r, d, g, lane); }
        else { goLeafBody<CPL>(v, pv, r, d, g, lane, smem); }
        waveSync();
    }
}

// bit-packed planes [B][C][W32] -> f32 planes [B][C][P] (the layout of env.getFeatures())
__global__ __launch_bounds__(256) void expand_bits_kernel(const uint32_t* __restrict__ bits, int C, int P, int W32, float* __restrict__ out)
{
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < C * P; i += 256) {
        const int c = i / P, p = i - c * P;
        out[size_t(b) * C * P + i] = ((bits[(size_t(b) * C + c) * W32 + (p >> 5)] >> (p & 31)) & 1) ? 1.0f : 0.0f;
    }
}

// Atari-shaped samples (ref atari.cpp:199-221): meta[b] = 8 pointers to screens of 3 x 96 x 96 bytes in device memory (oldest first), then 8 f32
// action-plane values, then 8 valid flags; planes: for each step [action_id / 18 everywhere][R][G][B] / 255 (invalid screen: zeros)
__global__ __launch_bounds__(256) void expand_atari_kernel(const uint8_t* __restrict__ meta, float* __restrict__ out)
{
    constexpr int kRes2 = 96 * 96, kFrame = 3 * kRes2, kHist = 8;
    const int b = blockIdx.y, step = blockIdx.x;
    const uint8_t* m = meta + size_t(b) * kAtariMetaBytes;
    uint64_t fp;
    float av;
    memcpy(&fp, m + step * 8, 8);
    memcpy(&av, m + 64 + step * 4, 4);
    const bool valid = m[96 + step] != 0;
    float* o = out + (size_t(b) * kHist + step) * 4 * kRes2;
    for (int p = threadIdx.x; p < kRes2; p += 256) { o[p] = av; }
    const uint8_t* f = reinterpret_cast<const uint8_t*>(fp);
    for (int i = threadIdx.x; i < kFrame; i += 256) { o[kRes2 + i] = valid ? static_cast<float>(f[i]) / 255.0f : 0.0f; }
}

} // namespace

int loaderReplayFeatures(GoDevice& gd, const PoolView& pv, int B, const int* d_pos, const uint8_t* d_rot, float* d_out, hipStream_t stream)
{
    const GoDevView& v = gd.v_;
    if (v.kind == 2) { hipLaunchKernelGGL((replay_kernel<2, 1>), dim3(B), dim3(64), 0, stream, v, pv, d_pos, d_rot); }
    else if (v.kind == 1) { hipLaunchKernelGGL((replay_kernel<1, 1>), dim3(B), dim3(64), 0, stream, v, pv, d_pos, d_rot); }
    else {
        const size_t smem = goLeafSmemBytes(v, pv.max_depth);
#define MZ_REPLAY_CASE(K) \
    case K: hipLaunchKernelGGL((replay_kernel<0, K>), dim3(B), dim3(64), smem, stream, v, pv, d_pos, d_rot); break;
        switch (v.W) {
            MZ_REPLAY_CASE(1) MZ_REPLAY_CASE(2) MZ_REPLAY_CASE(3) MZ_REPLAY_CASE(4) MZ_REPLAY_CASE(5) MZ_REPLAY_CASE(6)
        default: setError("loader: board too large for the device engine"); return MZ_ERR_ARG;
        }
#undef MZ_REPLAY_CASE
    }
    MZ_HIP(hipGetLastError());
    hipLaunchKernelGGL(expand_bits_kernel, dim3(B), dim3(256), 0, stream, v.feat, v.channels, v.P, v.W32, d_out);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

int loaderExpandAtari(const uint8_t* d_meta, int B, float* d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(expand_atari_kernel, dim3(8, B), dim3(256), 0, stream, d_meta, d_out);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

} // namespace mz
