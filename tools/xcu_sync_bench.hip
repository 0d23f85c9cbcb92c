// Micro-benchmark: cost of a per-layer exchange between the 4 workgroups that would share one game (one oc-tile of a 6x6x64 layer each):
// every workgroup stores its 16 x 36 floats, the four meet at a counter in global memory, every workgroup reads the other three parts.
// Variants: (0) agent-scope release/acquire atomics as the compiler emits them (L2 write-back + invalidate on a multi-XCD part),
// (1) relaxed atomics at L2 + loads that bypass the CU's vector cache — only valid when the four workgroups share an XCD (one L2); the
// kernel reads XCC_ID to check that workgroup ids congruent mod 8 do land on one XCD; (2) 8-byte (value, sequence) words polled by every
// thread at once; (3) 16-byte words (three values + sequence) polled after a short delay with a pause between polls; (4) the counter of (1)
// with every wave arriving and polling for itself.
// build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/xcu_sync_bench tools/xcu_sync_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kMembers = 4, kPart = 16 * 36, kThreads = 512;

struct Args {
    float* xchg;        // [clusters][2][kMembers][kPart] (variant 2: (value, sequence) pairs, twice the size)
    unsigned* counter;  // [clusters] monotonically increasing arrivals
    unsigned* xcc;      // [workgroups]
    unsigned long long* ticks; // [workgroups]
    float* sink;        // [workgroups]
    int clusters, iters, variant;
};

__device__ __forceinline__ float loadBypass(const float* p)
{
    float v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__global__ __launch_bounds__(kThreads) void bench(Args a)
{
    const int wg = blockIdx.x, cluster = wg % a.clusters, member = wg / a.clusters, tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[wg] = id;
    }
    __shared__ float lds[kMembers * kPart];
    float acc = 0.0f;
    unsigned* cnt = a.counter + cluster * 32; // one 128-byte line per cluster
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < a.iters; ++it) {
        float* x = a.xchg + (size_t(cluster) * 2 + (it & 1)) * kMembers * kPart;
        // my part: values that depend on (iteration, member, index) so that a stale read shows up in the checksum
        if (a.variant != 2 && a.variant != 3) { for (int i = tid; i < kPart; i += kThreads) { x[member * kPart + i] = float((it * 7 + member * 3 + i) & 1023); } }
        if (a.variant == 0) {
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = unsigned(it + 1) * kMembers;
                for (int polls = 0; polls < 1000000 && __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want; ++polls) { __builtin_amdgcn_s_sleep(1); }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int i = tid; i < kMembers * kPart; i += kThreads) { lds[i] = x[i]; }
        } else if (a.variant == 2) { // every value travels with the iteration number: the readers poll the data itself (no counter round trip)
            uint2* xx = reinterpret_cast<uint2*>(a.xchg) + (size_t(cluster) * 2 + (it & 1)) * kMembers * kPart;
            const unsigned seq = unsigned(it + 1);
            for (int i = tid; i < kPart; i += kThreads) {
                const float v = float((it * 7 + member * 3 + i) & 1023);
                uint2 pr; pr.x = __float_as_uint(v); pr.y = seq;
                asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(xx + member * kPart + i), "v"(pr) : "memory");
            }
            constexpr int kPer = (kMembers * kPart + kThreads - 1) / kThreads;
            uint2 got[kPer];
            bool ok;
            int polls = 0;
            do {
#pragma unroll
                for (int k = 0; k < kPer; ++k) {
                    const int i = tid + k * kThreads;
                    const uint2* src = xx + (i < kMembers * kPart ? i : 0);
                    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(got[k]) : "v"(src) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ok = true;
#pragma unroll
                for (int k = 0; k < kPer; ++k) { ok = ok && (got[k].y == seq || tid + k * kThreads >= kMembers * kPart); }
            } while (!__all(ok) && ++polls < 100000);
            if (polls >= 100000) { acc += 1.0f; }
#pragma unroll
            for (int k = 0; k < kPer; ++k) { const int i = tid + k * kThreads; if (i < kMembers * kPart) { lds[i] = __uint_as_float(got[k].x); } }
        } else if (a.variant == 3) { // 16-byte words: three values + the iteration number; the readers wait a little before the first poll
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4* xx = reinterpret_cast<u4*>(a.xchg) + (size_t(cluster) * 2 + (it & 1)) * kMembers * (kPart / 3);
            const unsigned seq = unsigned(it + 1);
            constexpr int kW = kPart / 3; // words per member
            for (int i = tid; i < kW; i += kThreads) {
                u4 w;
                w.x = __float_as_uint(float((it * 7 + member * 3 + 3 * i) & 1023));
                w.y = __float_as_uint(float((it * 7 + member * 3 + 3 * i + 1) & 1023));
                w.z = __float_as_uint(float((it * 7 + member * 3 + 3 * i + 2) & 1023));
                w.w = seq;
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(xx + member * kW + i), "v"(w) : "memory");
            }
            constexpr int kPer = (kMembers * kW + kThreads - 1) / kThreads;
            u4 got[kPer];
            bool ok;
            int polls = 0;
            __builtin_amdgcn_s_sleep(8);
            do {
#pragma unroll
                for (int k = 0; k < kPer; ++k) {
                    const int i = tid + k * kThreads;
                    const u4* src = xx + (i < kMembers * kW ? i : 0);
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(got[k]) : "v"(src) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ok = true;
#pragma unroll
                for (int k = 0; k < kPer; ++k) { ok = ok && (got[k].w == seq || tid + k * kThreads >= kMembers * kW); }
                if (!__all(ok)) { __builtin_amdgcn_s_sleep(2); }
            } while (!__all(ok) && ++polls < 100000);
            if (polls >= 100000) { acc += 1.0f; }
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int i = tid + k * kThreads;
                if (i < kMembers * kW) { lds[3 * i] = __uint_as_float(got[k].x); lds[3 * i + 1] = __uint_as_float(got[k].y); lds[3 * i + 2] = __uint_as_float(got[k].z); }
            }
        } else if (a.variant == 4) { // counter protocol, every wave arrives and polls for itself (no barrier before / after the rendezvous)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int nw = kThreads / 64;
            if ((tid & 63) == 0) { __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            const unsigned want = unsigned(it + 1) * kMembers * nw;
            for (int polls = 0; polls < 1000000 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++polls) { __builtin_amdgcn_s_sleep(1); }
            constexpr int kPer = (kMembers * kPart + kThreads - 1) / kThreads;
            float got[kPer];
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int i = tid + k * kThreads;
                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(got[k]) : "v"(x + (i < kMembers * kPart ? i : 0)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < kPer; ++k) { const int i = tid + k * kThreads; if (i < kMembers * kPart) { lds[i] = got[k]; } }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my stores have reached the L2
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = unsigned(it + 1) * kMembers;
                for (int polls = 0; polls < 1000000 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++polls) { __builtin_amdgcn_s_sleep(1); }
            }
            __syncthreads();
            constexpr int kPer = (kMembers * kPart + kThreads - 1) / kThreads;
            float got[kPer];
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int i = tid + k * kThreads;
                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(got[k]) : "v"(x + (i < kMembers * kPart ? i : 0)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < kPer; ++k) { const int i = tid + k * kThreads; if (i < kMembers * kPart) { lds[i] = got[k]; } }
        }
        __syncthreads();
        // checksum of what the other members wrote
        for (int i = tid; i < kMembers * kPart; i += kThreads) {
            const int m = i / kPart, j = i - m * kPart;
            acc += lds[i] - float((it * 7 + m * 3 + j) & 1023);
        }
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) { a.ticks[wg] = t1 - t0; }
    a.sink[wg * kThreads + tid] = acc;
}

int main(int argc, char** argv)
{
    const int clusters = argc > 1 ? atoi(argv[1]) : 64, iters = argc > 2 ? atoi(argv[2]) : 2000;
    const int wgs = clusters * kMembers;
    Args a{};
    a.clusters = clusters; a.iters = iters;
    CK(hipMalloc(&a.xchg, size_t(clusters) * 2 * kMembers * kPart * 8));
    CK(hipMemset(a.xchg, 0, size_t(clusters) * 2 * kMembers * kPart * 8));
    CK(hipMalloc(&a.counter, clusters * 32 * 4));
    CK(hipMalloc(&a.xcc, wgs * 4));
    CK(hipMalloc(&a.ticks, wgs * 8));
    CK(hipMalloc(&a.sink, size_t(wgs) * kThreads * 4));
    int coop = 0;
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bench, kThreads, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("cooperative launch %d, blocks per CU %d, CUs %d\n", coop, per_cu, prop.multiProcessorCount);
    for (int variant = 0; variant < 5; ++variant) {
        a.variant = variant;
        CK(hipMemset(a.counter, 0, clusters * 32 * 4));
        void* params[] = {&a};
        CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(bench), dim3(wgs), dim3(kThreads), params, 0, nullptr));
        CK(hipDeviceSynchronize());
        std::vector<unsigned> xcc(wgs); std::vector<unsigned long long> ticks(wgs); std::vector<float> sink(size_t(wgs) * kThreads);
        CK(hipMemcpy(xcc.data(), a.xcc, wgs * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ticks.data(), a.ticks, wgs * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sink.data(), a.sink, sink.size() * 4, hipMemcpyDeviceToHost));
        double bad = 0; for (float v : sink) { bad += v != 0.0f; }
        unsigned long long mx = 0; for (auto t : ticks) { mx = t > mx ? t : mx; }
        int split = 0;
        for (int c = 0; c < clusters; ++c) { for (int m = 1; m < kMembers; ++m) { split += (xcc[m * clusters + c] & 15) != (xcc[c] & 15); } }
        printf("variant %d: %.3f us per exchange (100-MHz clock), stale/incorrect lanes %.0f, cluster members on another XCD than member 0: %d\n", variant,
               double(mx) / 100.0 / iters, bad, split);
        if (variant == 0) { printf("xcc of wg 0..15:"); for (int i = 0; i < 16 && i < wgs; ++i) { printf(" %u", xcc[i] & 15); } printf("\n"); }
    }
    return 0;
}
