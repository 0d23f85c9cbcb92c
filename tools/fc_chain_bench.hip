// Micro-benchmark: how fast can one CU run the ordered f32 chain of a fully connected layer out of LDS?  acc = fma(x[i], w[i][lane], acc), i < n,
// with w rows of `seg` floats in LDS (the ring of fcStreamSeg) and x a vector shared by the wave.  Variants differ in how x reaches the fma
// (scalar operand via v_readlane, per-lane 16-byte LDS reads, broadcast 4-byte LDS reads), in whether the row stride is a compile-time
// constant (immediate DS offsets) and in whether the LDS reads of the next group are issued before the fmas of the current one.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/fc_chain_bench tools/fc_chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 96, kSeg = 64; // one chunk: 96 rows x 64 floats = 24 KB

template <int V, int G>
__device__ __forceinline__ float chain(const float* __restrict__ x, const float* __restrict__ rs, int seg, int rows, int lane)
{
    float acc = 0.0f;
    if constexpr (V == 4) { // plain loop
        for (int r = 0; r < rows; ++r) { acc = __builtin_fmaf(x[r], rs[r * seg + lane], acc); }
    } else if constexpr (V == 0) { // v_readlane, run-time stride
        for (int r0 = 0; r0 + G <= rows; r0 += G) {
            const float xl = x[r0 + (lane < G ? lane : 0)];
            float wv[G];
#pragma unroll
            for (int r = 0; r < G; ++r) { wv[r] = rs[(r0 + r) * seg + lane]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < G; ++r) { acc = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xl), r)), wv[r], acc); }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (V == 1) { // per-lane 16-byte reads of x, run-time stride
        for (int r0 = 0; r0 + G <= rows; r0 += G) {
            vf4 xv[G / 4];
            float wv[G];
#pragma unroll
            for (int r = 0; r < G; r += 4) { xv[r / 4] = *reinterpret_cast<const vf4*>(x + r0 + r); }
#pragma unroll
            for (int r = 0; r < G; ++r) { wv[r] = rs[(r0 + r) * seg + lane]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < G; r += 4) {
                acc = __builtin_fmaf(xv[r / 4].x, wv[r], acc); acc = __builtin_fmaf(xv[r / 4].y, wv[r + 1], acc);
                acc = __builtin_fmaf(xv[r / 4].z, wv[r + 2], acc); acc = __builtin_fmaf(xv[r / 4].w, wv[r + 3], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (V == 2) { // per-lane 16-byte reads of x, compile-time stride (immediate offsets)
        for (int r0 = 0; r0 + G <= rows; r0 += G) {
            vf4 xv[G / 4];
            float wv[G];
            const float* rp = rs + r0 * kSeg + lane;
#pragma unroll
            for (int r = 0; r < G; r += 4) { xv[r / 4] = *reinterpret_cast<const vf4*>(x + r0 + r); }
#pragma unroll
            for (int r = 0; r < G; ++r) { wv[r] = rp[r * kSeg]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < G; r += 4) {
                acc = __builtin_fmaf(xv[r / 4].x, wv[r], acc); acc = __builtin_fmaf(xv[r / 4].y, wv[r + 1], acc);
                acc = __builtin_fmaf(xv[r / 4].z, wv[r + 2], acc); acc = __builtin_fmaf(xv[r / 4].w, wv[r + 3], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (V == 3) { // V2 + the next group's reads are issued before this group's fmas
        vf4 xa[G / 4], xb[G / 4];
        float wa[G], wb[G];
        auto rd = [&](vf4 (&xv)[G / 4], float (&wv)[G], int r0) {
            const float* rp = rs + r0 * kSeg + lane;
#pragma unroll
            for (int r = 0; r < G; r += 4) { xv[r / 4] = *reinterpret_cast<const vf4*>(x + r0 + r); }
#pragma unroll
            for (int r = 0; r < G; ++r) { wv[r] = rp[r * kSeg]; }
        };
        auto fm = [&](const vf4 (&xv)[G / 4], const float (&wv)[G]) {
#pragma unroll
            for (int r = 0; r < G; r += 4) {
                acc = __builtin_fmaf(xv[r / 4].x, wv[r], acc); acc = __builtin_fmaf(xv[r / 4].y, wv[r + 1], acc);
                acc = __builtin_fmaf(xv[r / 4].z, wv[r + 2], acc); acc = __builtin_fmaf(xv[r / 4].w, wv[r + 3], acc);
            }
        };
        rd(xa, wa, 0);
        for (int r0 = 0; r0 + 2 * G <= rows; r0 += 2 * G) {
            rd(xb, wb, r0 + G);
            __builtin_amdgcn_sched_barrier(0);
            fm(xa, wa);
            __builtin_amdgcn_sched_barrier(0);
            rd(xa, wa, r0 + 2 * G < rows ? r0 + 2 * G : 0);
            __builtin_amdgcn_sched_barrier(0);
            fm(xb, wb);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (V == 5) { // broadcast 4-byte reads of x (uniform address), compile-time stride
        for (int r0 = 0; r0 + G <= rows; r0 += G) {
            float xv[G], wv[G];
            const float* rp = rs + r0 * kSeg + lane;
#pragma unroll
            for (int r = 0; r < G; ++r) { xv[r] = x[r0 + r]; wv[r] = rp[r * kSeg]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < G; ++r) { acc = __builtin_fmaf(xv[r], wv[r], acc); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

template <int V, int G>
__global__ __launch_bounds__(512) void bench(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ ticks, int seg, int rows, int reps,
                                             int waves)
{
    __shared__ __attribute__((aligned(16))) float ring[kRows * kSeg];
    __shared__ __attribute__((aligned(16))) float xs[4][kRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kRows * kSeg; i += 512) { ring[i] = src[i]; }
    for (int i = tid; i < 4 * kRows; i += 512) { xs[i / kRows][i % kRows] = src[kRows * kSeg + i]; }
    __syncthreads();
    float acc = 0.0f;
    const unsigned long long t0 = wall_clock64();
    if (wave < waves) {
        for (int it = 0; it < reps; ++it) { acc += chain<V, G>(xs[wave & 3], ring, seg, rows, lane); }
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x * 512 + tid] = acc;
    if (tid == 0) { ticks[blockIdx.x] = t1 - t0; }
}

template <int V, int G>
static void run(const char* name, const float* d_src, float* d_out, unsigned long long* d_ticks, int waves)
{
    const int reps = 200, blocks = 64;
    hipLaunchKernelGGL((bench<V, G>), dim3(blocks), dim3(512), 0, nullptr, d_src, d_out, d_ticks, kSeg, kRows, reps, waves);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> t(blocks);
    CK(hipMemcpy(t.data(), d_ticks, blocks * 8, hipMemcpyDeviceToHost));
    std::vector<float> o(512);
    CK(hipMemcpy(o.data(), d_out, 512 * 4, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : t) { avg += double(v); } avg /= blocks;
    printf("%-52s waves %d: %.2f ns per row (%.1f cycles at 2.4 GHz), checksum %.6g\n", name, waves, avg * 10.0 / (double(reps) * kRows), avg * 10.0 / (double(reps) * kRows) * 2.4,
           double(o[5]));
}

int main()
{
    const size_t n = kRows * kSeg + 4 * kRows;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) { h[i] = float((i * 2654435761u) % 1000) / 1000.0f - 0.5f; }
    float *d_src, *d_out; unsigned long long* d_ticks;
    CK(hipMalloc(&d_src, n * 4)); CK(hipMalloc(&d_out, 64 * 512 * 4)); CK(hipMalloc(&d_ticks, 64 * 8));
    CK(hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice));
    for (int waves : {1, 4, 8}) {
        run<4, 24>("plain loop", d_src, d_out, d_ticks, waves);
        run<0, 24>("readlane x, run-time stride, groups of 24", d_src, d_out, d_ticks, waves);
        run<1, 24>("16-byte x, run-time stride, groups of 24", d_src, d_out, d_ticks, waves);
        run<2, 24>("16-byte x, immediate offsets, groups of 24", d_src, d_out, d_ticks, waves);
        run<3, 24>("16-byte x, immediate offsets, pipelined groups of 24", d_src, d_out, d_ticks, waves);
        run<3, 48>("16-byte x, immediate offsets, pipelined groups of 48", d_src, d_out, d_ticks, waves);
        run<5, 24>("4-byte broadcast x, immediate offsets, groups of 24", d_src, d_out, d_ticks, waves);
    }
    return 0;
}
