// Device-side helpers shared by the network kernels (net.hip, net_atari.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS tiles of the fused tower: the blocks' input/output x (the second conv writes y over x: every lane reads the skip value and writes
// the result at ITS OWN (channel, pixel), the conv itself reads the temporary) and the temporary (also the stem's input)
constexpr int kTowerTiles = 2;

__host__ __device__ constexpr int planeStride(int H, int W)
{
    // padded plane (H+2)*(W+2) rounded up so that stride % 32 == 16: the two 16-lane channel groups
    // that share a 32-lane ds_read_b32 service group then start on different bank halves
    int ps = (H + 2) * (W + 2);
    int r = ps % 32;
    return ps + ((16 - r) + 32) % 32;
}

// ---------------------------------------------------------------------------------------------
// deterministic exp / tanh (same operation sequence as the CPU oracle)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float mz_expf(float x)
{
    if (x < -87.0f) { return 0.0f; }
    if (x > 88.0f) { x = 88.0f; }
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    int ni = static_cast<int>(n);
    float scale = __builtin_bit_cast(float, static_cast<unsigned>(ni + 127) << 23);
    return y * scale;
}
__device__ __forceinline__ float mz_tanhf(float x)
{
    float ax = __builtin_fabsf(x);
    if (ax > 10.0f) { return __builtin_copysignf(1.0f, x); }
    float e = mz_expf(-2.0f * ax);
    float t = (1.0f - e) / (1.0f + e);
    return __builtin_copysignf(t, x);
}

} // namespace mz
