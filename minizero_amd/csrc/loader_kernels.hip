// Device side of the learner-side sampler (loader.cpp): the feature planes of a sampled (game, position) are produced ON the GPU, straight into
// the training batch in HBM.  The reference replays every sampled game from its first move on a fresh host environment
// (ref environment/base/base_env.h:235-241 BaseEnvLoader::getFeatures, called per sample by learner/data_loader.cpp:141-151,163-170).
// Here one wave64 per sample replays its game on the device rules engine the self-play worker already has (go_body.h: one position slot per
// move, leaf = parent slot + one move), for all samples of the batch at once; Atari-shaped samples expand their 8 stored screens.
#include "go_body.h"
#include "loader_dev.h"

namespace mz {

namespace {

// sample g: moves pact[g][1 .. pos[g]] replayed from the root snapshot (slot 0); the planes of the last position under rot[g] end in v.feat
template <int KIND, int CPL>
__global__ __launch_bounds__(64) void replay_kernel(GoDevView v, PoolView pv, const int* __restrict__ pos, const uint8_t* __restrict__ rot)
{
    extern __shared__ uint64_t smem[];
    const int g = blockIdx.x, lane = threadIdx.x;
    const int n = pos[g], r = rot[g];
    for (int d = 0; d <= n; ++d) {
        if (lane == 0) { pv.path_len[g] = d + 1; }
        waveSync();
        if (d == 0 && n > 0) { continue; } // slot 0 is the uploaded root: only evaluate it when it is the sampled position itself
        if constexpr (KIND == 2) { tttLeafBody(v, pv, r, d, g, lane); }
        else if constexpr (KIND == 1) { othLeafBody(v, pv, r, d, g, lane); }
        else { goLeafBody<CPL>(v, pv, r, d, g, lane, smem, nullptr, d < n); }
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

// Atari-shaped samples (ref atari.cpp:199-221): raw[b] = 8 screens of 3 x 96 x 96 bytes (oldest first), then 8 f32 action-plane values, then
// 8 valid flags; planes: for each step [action_id / 18 everywhere][R][G][B] / 255 (invalid screen: zeros)
__global__ __launch_bounds__(256) void expand_atari_kernel(const uint8_t* __restrict__ raw, int raw_bytes, float* __restrict__ out)
{
    constexpr int kRes2 = 96 * 96, kFrame = 3 * kRes2, kHist = 8;
    const int b = blockIdx.y, step = blockIdx.x;
    const uint8_t* r = raw + size_t(b) * raw_bytes;
    float av;
    memcpy(&av, r + size_t(kHist) * kFrame + step * 4, 4);
    const bool valid = r[size_t(kHist) * kFrame + kHist * 4 + step] != 0;
    float* o = out + (size_t(b) * kHist + step) * 4 * kRes2;
    for (int p = threadIdx.x; p < kRes2; p += 256) { o[p] = av; }
    const uint8_t* f = r + size_t(step) * kFrame;
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

int loaderExpandAtari(const uint8_t* d_raw, int raw_bytes, int B, float* d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(expand_atari_kernel, dim3(8, B), dim3(256), 0, stream, d_raw, raw_bytes, d_out);
    MZ_HIP(hipGetLastError());
    return MZ_OK;
}

} // namespace mz
