// What the MFMA pipe of one SIMD sustains under the tower's instruction mix (tools only; build: hipcc --offload-arch=gfx950 -O3 tools/mfma_issue.hip -o ab/mfma_issue).
// Two waves per SIMD (512 threads per workgroup, one workgroup per CU) issue v_mfma_f32_16x16x4_f32 on NT independent accumulators; the variants add, step by
// step, what a layer of the fused tower does around them.  Prints shader cycles per MFMA and SIMD (the pipe's own rate is 32).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int MODE> // MODE 0: MFMAs only; 1: + one ds_read_b32 per MFMA, a group ahead; 2: + 16 A fragments (global, 4 x dwordx4) every 16 groups, a block ahead
__global__ __launch_bounds__(512) void mfma_issue(const float* __restrict__ w, float* __restrict__ out, unsigned long long* cyc, int groups)
{
    __shared__ float tile[64 * 144 * 2];
    for (int i = threadIdx.x; i < 64 * 144 * 2; i += 512) { tile[i] = float(i & 7) * 0.125f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc[j] = f32x4{0, 0, 0, 0}; }
    int off[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { off[j] = (lane >> 4) * 144 + (wave >> 2) * 48 + j * 16 + (lane & 15) + ((lane & 15) >= 9 ? 2 : 0); }
    float a[16], an[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = 0.5f + float(i + lane) * 0.001f; an[i] = a[i]; }
    const float* wl = w + size_t(wave & 3) * 16 * 64 + lane * 4;
    float bc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { bc[j] = tile[off[j]]; }
    const unsigned long long c0 = clock64();
#pragma unroll 1
    for (int g0 = 0; g0 < groups; g0 += 16) {
        if (MODE >= 2) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float4 v = *reinterpret_cast<const float4*>(wl + size_t((g0 >> 4) % 9) * 4 * 16 * 64 + c4 * 256);
                an[4 * c4] = v.x; an[4 * c4 + 1] = v.y; an[4 * c4 + 2] = v.z; an[4 * c4 + 3] = v.w;
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int cg = 0; cg < 16; ++cg) {
            float bn[NT];
            if (MODE >= 1) {
#pragma unroll
                for (int j = 0; j < NT; ++j) { bn[j] = tile[off[j] + ((cg + 1) & 15) * 4 * 144 + ((g0 >> 4) % 3)]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cg], bc[j], acc[j], 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1) {
#pragma unroll
                for (int j = 0; j < NT; ++j) { bc[j] = bn[j]; }
            }
        }
        if (MODE >= 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { a[i] = an[i]; }
        }
    }
    const unsigned long long c1 = clock64();
    float s = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) { s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3]; }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) { cyc[blockIdx.x * 8 + wave] = c1 - c0; }
}

template <int NT, int MODE>
void run(const float* w, float* out, unsigned long long* cyc, const char* what)
{
    const int groups = 16 * 9 * 13;
    for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL((mfma_issue<NT, MODE>), dim3(256), dim3(512), 0, 0, w, out, cyc, groups); }
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; unsigned long long mx = 0;
    for (int i = 0; i < 256 * 8; ++i) { sum += double(h[i]); if (h[i] > mx) mx = h[i]; }
    const double per = double(groups) * NT * 2; // MFMAs per SIMD (two waves)
    printf("NT=%d %-46s %6.2f cycles per MFMA and SIMD (mean over waves), %6.2f (slowest wave)\n", NT, what, sum / (256 * 8) / per, double(mx) / per);
}

// variants of MODE 1: V 0 = as the tower (one ds_read_b32 per MFMA, a k-group ahead, pinned order); 1 = conflict-free addresses; 2 = two k-groups ahead;
// 3 = one ds_read_b128 per four k-groups and tile (a quad ahead); 4 = as 0 without the scheduling fences; 5 = one ds_read_b64 per two k-groups
// V 6 / 7: the real addresses of a 9x9 board (plane stride 144, row stride 11): 16 consecutive pixels per tile (the tower's TileMap), and tiles chosen so that the 64
// lanes of a read fall on 64 different banks (four of the five full tiles; a fifth cannot: the padded positions are not uniform modulo 16)
__constant__ int kTilesRowMajor[6][16] = {{0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15},
    {16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31},
    {32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47},
    {48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63},
    {64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79},
    {80,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1}};
__constant__ int kTilesSpread[6][16] = {{4,5,12,16,25,32,40,45,46,49,62,65,67,69,73,76},
    {0,2,3,9,10,15,19,20,21,22,26,30,39,57,58,63},
    {1,6,7,13,14,23,24,35,37,44,48,55,60,68,71,78},
    {8,11,18,29,31,41,42,47,50,52,56,59,66,74,75,79},
    {17,27,28,33,34,36,38,43,51,53,54,61,64,70,72,77},
    {80,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1}};
template <int NT, int V>
__global__ __launch_bounds__(512) void mfma_lds(float* __restrict__ out, unsigned long long* cyc, int groups)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * 144 * 2];
    for (int i = threadIdx.x; i < 64 * 144 * 2; i += 512) { tile[i] = float(i & 7) * 0.125f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc[j] = f32x4{0, 0, 0, 0}; }
    int off[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        off[j] = (lane >> 4) * 144 + (wave >> 2) * 48 + j * 16 + (lane & 15) + ((lane & 15) >= 9 ? 2 : 0);
        if (V == 1) { off[j] = lane + j * 64 + (wave >> 2) * 256; }
        if (V == 3) { off[j] = ((lane >> 4) * 144 + (wave >> 2) * 24 + j * 8 + (lane & 15) / 2) * 4; } // 16-byte units: 4 consecutive floats per (plane, pixel)
        if (V == 5) { off[j] = ((lane >> 4) * 144 + (wave >> 2) * 48 + j * 16 + (lane & 15)) * 2; }
        if (V == 6 || V == 7) {
            int q = (V == 6 ? kTilesRowMajor : kTilesSpread)[(wave >> 2) * 3 + j][lane & 15];
            q = q < 0 ? 0 : q;
            off[j] = (lane >> 4) * 144 + (q / 9) * 11 + q % 9;
        }
    }
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = 0.5f + float(i + lane) * 0.001f; }
    const unsigned long long c0 = clock64();
    if constexpr (V == 3) {
        float4 q[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { q[j] = *reinterpret_cast<const float4*>(tile + off[j]); }
#pragma unroll 1
        for (int g0 = 0; g0 < groups; g0 += 16) {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                float4 qn[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) { qn[j] = *reinterpret_cast<const float4*>(tile + off[j] + ((cq + 1) & 3) * 4 * 144 * 4 + ((g0 >> 4) % 3) * 4); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float b = k == 0 ? q[j].x : k == 1 ? q[j].y : k == 2 ? q[j].z : q[j].w;
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * cq + k], b, acc[j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NT; ++j) { q[j] = qn[j]; }
            }
        }
    } else if constexpr (V == 5) {
        float2 q[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { q[j] = *reinterpret_cast<const float2*>(tile + off[j]); }
#pragma unroll 1
        for (int g0 = 0; g0 < groups; g0 += 16) {
#pragma unroll
            for (int cq = 0; cq < 8; ++cq) {
                float2 qn[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) { qn[j] = *reinterpret_cast<const float2*>(tile + off[j] + ((cq + 1) & 3) * 4 * 144 * 2 + ((g0 >> 4) % 3) * 2); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * cq + k], k == 0 ? q[j].x : q[j].y, acc[j], 0, 0, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NT; ++j) { q[j] = qn[j]; }
            }
        }
    } else {
        float bc[NT], bn[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { bc[j] = tile[off[j]]; bn[j] = tile[off[j] + 4 * 144]; }
#pragma unroll 1
        for (int g0 = 0; g0 < groups; g0 += 16) {
#pragma unroll
            for (int cg = 0; cg < 16; ++cg) {
                float b2[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) { b2[j] = tile[off[j] + ((cg + (V == 2 ? 2 : 1)) & 15) * 4 * (V == 1 ? 128 : 144) + ((g0 >> 4) % 3)]; }
                if (V != 4) { __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int j = 0; j < NT; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cg], bc[j], acc[j], 0, 0, 0); }
                if (V != 4) { __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int j = 0; j < NT; ++j) { bc[j] = V == 2 ? bn[j] : b2[j]; bn[j] = b2[j]; }
            }
        }
    }
    const unsigned long long c1 = clock64();
    float s = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) { s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3]; }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) { cyc[blockIdx.x * 8 + wave] = c1 - c0; }
}

template <int NT, int V>
void runLds(float* out, unsigned long long* cyc, const char* what)
{
    const int groups = 16 * 9 * 13;
    for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL((mfma_lds<NT, V>), dim3(256), dim3(512), 0, 0, out, cyc, groups); }
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; unsigned long long mx = 0;
    for (int i = 0; i < 256 * 8; ++i) { sum += double(h[i]); if (h[i] > mx) mx = h[i]; }
    const double per = double(groups) * NT * 2;
    printf("NT=%d %-46s %6.2f cycles per MFMA and SIMD (mean over waves), %6.2f (slowest wave)\n", NT, what, sum / (256 * 8) / per, double(mx) / per);
}

int main()
{
    float *w, *out; unsigned long long* cyc;
    hipMalloc(&w, size_t(9) * 4 * 16 * 64 * 4 + 4096); hipMemset(w, 0, size_t(9) * 4 * 16 * 64 * 4 + 4096);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    run<3, 0>(w, out, cyc, "MFMAs only");
    run<3, 1>(w, out, cyc, "+ B operand from LDS, a k-group ahead");
    run<3, 2>(w, out, cyc, "+ A fragments from global memory, a tap ahead");
    run<2, 0>(w, out, cyc, "MFMAs only");
    run<2, 1>(w, out, cyc, "+ B operand from LDS, a k-group ahead");
    run<2, 2>(w, out, cyc, "+ A fragments from global memory, a tap ahead");
    run<4, 1>(w, out, cyc, "+ B operand from LDS, a k-group ahead");
    run<6, 1>(w, out, cyc, "+ B operand from LDS, a k-group ahead");
    runLds<3, 0>(out, cyc, "LDS b32 a group ahead (the tower's)");
    runLds<3, 1>(out, cyc, "  conflict-free addresses");
    runLds<3, 2>(out, cyc, "  two groups ahead");
    runLds<3, 4>(out, cyc, "  without scheduling fences");
    runLds<3, 5>(out, cyc, "  ds_read_b64 per two groups");
    runLds<3, 3>(out, cyc, "  ds_read_b128 per four groups");
    runLds<3, 6>(out, cyc, "  9x9 addresses, tiles of 16 consecutive pixels");
    runLds<3, 7>(out, cyc, "  9x9 addresses, tiles spread over the banks");
    runLds<2, 0>(out, cyc, "LDS b32 a group ahead (the tower's)");
    runLds<2, 3>(out, cyc, "  ds_read_b128 per four groups");
    return 0;
}
