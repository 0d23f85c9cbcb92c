// Time stamps of the tower's wave phases (tools only; build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I minizero_amd/csrc -I include
// tools/tower_prof.hip -o gpurun_out/tower_prof).  Prints, per layer and wave, the cycles from the layer's start to "all MFMAs issued",
// to "epilogue written" and to "barrier passed".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__shared__ unsigned long long s_tprof[8][64];
__shared__ int s_tprof_n[8];
#ifdef NO_TPROF // timing only (-DNO_TPROF): the stamps themselves cost ~3.5 k cycles per layer
#define MZ_TPROF(slot) do { } while (0)
#else
#define MZ_TPROF(slot)                                                                                   \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0) {                                                                   \
            const int w_ = threadIdx.x >> 6, i_ = s_tprof_n[w_];                                         \
            if (i_ < 64) { s_tprof[w_][i_] = (static_cast<unsigned long long>(clock64()) << 2) | (slot); } \
            s_tprof_n[w_] = i_ + 1;                                                                      \
        }                                                                                                \
    } while (0)
#endif
#include "net_dev.h"
#include "net_body.h"
using namespace mz;

template <int H, int W, int CIN0_PAD, int CPAD>
__global__ __launch_bounds__(512) void tower_prof(const float* in, const float* params, TowerArgs ta, float* out, unsigned long long* prof)
{
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    if ((threadIdx.x & 63) == 0) { s_tprof_n[threadIdx.x >> 6] = 0; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64(); // shader-clock cycles against the constant 100 MHz counter: the clock the kernel really ran at
    __syncthreads();
    towerBody<H, W, CIN0_PAD, CPAD>(in, params, ta, out, blockIdx.x, threadIdx.x, tiles);
    __syncthreads();
    if (blockIdx.x == 1 && threadIdx.x == 0) { prof[8 * 64] = clock64() - c0; prof[8 * 64 + 1] = wall_clock64() - w0; }
    if (blockIdx.x == 0 && threadIdx.x < 8) { for (int i = 0; i < 64; ++i) { prof[threadIdx.x * 64 + i] = i < s_tprof_n[threadIdx.x] ? s_tprof[threadIdx.x][i] : 0; } }
}

int main()
{
    constexpr int H = 9, W = 9, C0 = 20, C = 64, B = 256, NL = 13;
    TowerArgs ta{};
    ta.nlayers = NL; ta.cin0 = 18; ta.C = C; ta.OT = 4; ta.in_bits = 1; ta.has_stem = 1;
    size_t off = 0;
    for (int l = 0; l < NL; ++l) {
        const int cg = (l == 0 ? C0 : C) / 4;
        ta.w_off[l] = unsigned(off); off += size_t(9) * 4 * cg * 64;
        ta.b_off[l] = unsigned(off); off += 64;
    }
    std::vector<float> hp(off);
    for (size_t i = 0; i < off; ++i) { hp[i] = float((i * 2654435761u) % 1000) * 1e-4f - 0.05f; }
    float *params, *out; unsigned* in; unsigned long long* prof;
    hipMalloc(&params, off * 4); hipMemcpy(params, hp.data(), off * 4, hipMemcpyHostToDevice);
    hipMalloc(&in, size_t(B) * 18 * 3 * 4); hipMemset(in, 0x5a, size_t(B) * 18 * 3 * 4);
    hipMalloc(&out, size_t(B) * C * 81 * 4);
    hipMalloc(&prof, (8 * 64 + 2) * 8);
    const size_t lds = size_t(kTowerTiles) * 64 * planeStride(H, W) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(tower_prof<H, W, C0, C>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        for (int k = 0; k < 200; ++k) { hipLaunchKernelGGL((tower_prof<H, W, C0, C>), dim3(B), dim3(512), lds, 0, reinterpret_cast<const float*>(in), params, ta, out, prof); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch: %.1f us\n", ms * 1000 / 200);
    }
    std::vector<unsigned long long> h(8 * 64 + 2);
    hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
    printf("workgroup 1: %llu shader cycles in %llu ticks of the 100 MHz counter = %.0f MHz\n", h[512], h[513], double(h[512]) / double(h[513]) * 100.0);
    // per wave: sequence of (slot, clock): 0 start, 1 issued, 2 epilogue, 3 barrier
    for (int w = 0; w < 8; ++w) {
        printf("wave %d:", w);
        unsigned long long t0 = 0, prev3 = 0;
        double s_issue = 0, s_epi = 0, s_bar = 0, s_pre = 0; int n = 0;
        for (int i = 0; i < 64; ++i) {
            const unsigned long long v = h[w * 64 + i]; if (!v) break;
            const int slot = int(v & 3); const unsigned long long c = v >> 2;
            static unsigned long long t1, t2;
            if (slot == 0) { t0 = c; if (prev3) s_pre += double(c - prev3); }
            if (slot == 1) { t1 = c; }
            if (slot == 2) { t2 = c; }
            if (slot == 3) { s_issue += double(t1 - t0); s_epi += double(t2 - t1); s_bar += double(c - t2); prev3 = c; ++n; if (n <= 3 || n == 12) printf(" [L%d issue %llu epi %llu bar %llu]", n, t1 - t0, t2 - t1, c - t2); }
        }
        printf("\n   avg over %d layers: issue %.0f  epilogue %.0f  barrier-wait %.0f  pre %.0f cycles (100 MHz? see clock64 units)\n", n, s_issue / n, s_epi / n, s_bar / n, s_pre / std::max(1, n - 1));
    }
    return 0;
}
