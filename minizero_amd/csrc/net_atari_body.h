// Device bodies of the muzero_atari heads (ref network/py/muzero_atari_network.py:60-110,160-198, muzero_network.h:150-176), shared by the
// stand-alone kernel (net_atari.hip, 2 x 512 threads) and the per-game simulation kernel (sim.hip, 2 x 256 threads).
#pragma once
#include "net_dev.h"
#include "net_body.h"

#ifndef MZ_HPROF
#define MZ_HPROF(k) // experiment hook (sim.hip -DMZ_SIM_HPROF): time stamps inside the 601-bin heads
#endif

namespace mz {

struct DiscreteParams { const float *conv_w, *conv_b, *fc1_wT, *fc1_b, *fc2_wT, *fc2_b; int hc, hidden, size; };
struct AtariHeadParams {
    DiscreteParams reward, value;
    const float *pconv_w, *pconv_b, *pfc_wT, *pfc_b;
    int C, P, A, PC;
};

// DiscreteValueNetwork + softmax expectation on the LDS-resident activations xs[C][P]; result (transformed space) -> *out.
// Run by one HALF of the workgroup (512 threads, `t` = 0..511): the reward and the value head of a sample are independent and run
// side by side on the two halves.  Both halves pass the same barriers; `active` = false: barriers only.  Every sum is the reference's
// sequential f32 chain (dotChain: the weights of 16 steps are loaded ahead of the 16 dependent fmas); the 601 quotients of the
// expectation are independent and computed by all threads, only the two index-ordered sums are serial.
// (orderedSumWave: the index-ordered sum of an LDS vector by one wave, is in net_body.h)
// A slice of a fully connected layer with the weights STREAMED through LDS (sim_cluster.h: a 601-bin head alone on its CU, or a column slice of
// it for four games of an octet).  Chain k of thread t: acc[k] = the ordered f32 chain over i < n of xk[k][i] * W[i][col0 + uk[k]] (weights
// wT[n][ws] in global memory, uk[k] < seg; xk[k] = an LDS vector, 16-byte aligned).  ALL NT threads fetch chunk c + D - 1 (R rows x seg columns,
// D - 1 chunks in flight in registers) while chunk c is consumed from an LDS ring of two slots, so a layer does not pay the memory latency once per
// batch of rows.  VEC4: 16-byte loads (ws, col0, seg multiples of 4), else 4-byte loads.  G = rows per straight-line compute group (n % 4 == 0,
// G % 4 == 0).  RS > 0: seg == RS at compile time.  Ring = 2 * fcSegSlot floats, 16-byte aligned.
//
// What shaped this code (measured on BASELINE configs[4] and with tools/fc_chain_bench.hip):
//  * no branch around a global load: the compiler counts the loads in flight per PATH and waits for the oldest chunk with s_waitcnt vmcnt(N),
//    N = the loads it can prove to be younger; behind a conditional load N shrinks towards 0 and every chunk pays the full latency.  Chunks beyond
//    the layer in the last round therefore fetch element 0 and compute nothing;
//  * the weight pointer is cast to the global address space: through a generic pointer the loads are FLAT loads, which also count on lgkmcnt, so
//    every wait for an LDS operand waits for all weight loads in flight;
//  * rounds of D chunks form a real loop (loop-carried loads in flight do get exact vmcnt values): with the chunks fully unrolled — 58 copies of
//    the chunk code, each executed once per simulation — the heads ran out of the instruction cache;
//  * inside a group all LDS reads come first (sched_barrier), then the dependent fmas: left alone the scheduler puts every read right in front of
//    its fma, G LDS round trips in a row.  With a compile-time row stride the reads differ only in their immediate offsets: 14.7 instead of 22
//    cycles per row; handing x to the fmas as a scalar (v_readlane of a row-per-lane register) costs 33, a dependent fma alone ~8.
template <int NT, bool VEC4, int LPT>
__host__ __device__ constexpr int fcSegSlot() { return LPT * NT * (VEC4 ? 4 : 1); }
template <int NT, bool VEC4, int G, int LPT>
__host__ __device__ inline int fcSegRows(int seg) { const int r = fcSegSlot<NT, VEC4, LPT>() / seg / G * G; return r < G ? G : r; }

template <int NT, bool VEC4, int G, int K, int D, int LPT, int RS = 0>
__device__ __forceinline__ void fcStreamSeg(const float* const (&xk)[K], const int (&uk)[K], bool mine, const float* __restrict__ W, int ws, int col0, int seg_, int n,
                                            float* __restrict__ ring, int t, float (&acc)[K])
{
    const int seg = RS > 0 ? RS : seg_;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    typedef typename std::conditional<VEC4, vf4, float>::type LT;
    typedef __attribute__((address_space(1))) const LT GLT;
    static_assert(G % 4 == 0, "four rows per step");
    constexpr int SLOT = fcSegSlot<NT, VEC4, LPT>(), E = VEC4 ? 4 : 1;
    const int R = fcSegRows<NT, VEC4, G, LPT>(seg), nchunks = (n + R - 1) / R, per = seg / E; // load elements per row
    GLT* Wg = (GLT*)(W);
    LT buf[D][LPT];
    int rowj[LPT], offj[LPT]; // row within a chunk and element offset within the matrix of this thread's LPT loads (chunk 0)
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        const int e = t + j * NT;
        rowj[j] = e / per;
        offj[j] = (rowj[j] * ws + col0) / E + (e - rowj[j] * per);
        if (rowj[j] >= R) { rowj[j] = 1 << 28; } // beyond the chunk: never fetched (element 0 instead)
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { acc[k] = 0.0f; }
    const int cstep = R * ws / E; // element offset between consecutive chunks (R * ws is a multiple of 4 for VEC4: ws is)
    auto issue = [&](LT (&b)[LPT], int c) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const bool ok = c * R + rowj[j] < n;
            b[j] = Wg[ok ? c * cstep + offj[j] : 0];
        }
    };
    auto drain = [&](const LT (&b)[LPT], int c) {
        LT* dst = reinterpret_cast<LT*>(ring + size_t(c & 1) * SLOT);
#pragma unroll
        for (int j = 0; j < LPT; ++j) { dst[t + j * NT] = b[j]; }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) { issue(buf[d], d); }
    drain(buf[0], 0);
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int c = c0 + d;
            issue(buf[d], c + D);
            drain(buf[(d + 1) % D], c + 1);
            const int rows = n - c * R < R ? n - c * R : R;
            if (mine && rows > 0) {
                const float* rs = ring + size_t(c & 1) * SLOT;
                int r0 = 0;
                for (; r0 + G <= rows; r0 += G) {
                    vf4 xv[K][G / 4];
                    float wv[K][G];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float* rp = rs + r0 * seg + uk[k];
#pragma unroll
                        for (int r = 0; r < G; r += 4) { xv[k][r / 4] = *reinterpret_cast<const vf4*>(xk[k] + c * R + r0 + r); }
#pragma unroll
                        for (int r = 0; r < G; ++r) { wv[k][r] = rp[r * seg]; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        float a = acc[k];
#pragma unroll
                        for (int r = 0; r < G; r += 4) {
                            a = __builtin_fmaf(xv[k][r / 4].x, wv[k][r + 0], a);
                            a = __builtin_fmaf(xv[k][r / 4].y, wv[k][r + 1], a);
                            a = __builtin_fmaf(xv[k][r / 4].z, wv[k][r + 2], a);
                            a = __builtin_fmaf(xv[k][r / 4].w, wv[k][r + 3], a);
                        }
                        acc[k] = a;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (; r0 < rows; r0 += 4) { // the layer's tail (n % 4 == 0)
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const vf4 x4 = *reinterpret_cast<const vf4*>(xk[k] + c * R + r0);
                        float a = acc[k];
                        a = __builtin_fmaf(x4.x, rs[(r0 + 0) * seg + uk[k]], a);
                        a = __builtin_fmaf(x4.y, rs[(r0 + 1) * seg + uk[k]], a);
                        a = __builtin_fmaf(x4.z, rs[(r0 + 2) * seg + uk[k]], a);
                        a = __builtin_fmaf(x4.w, rs[(r0 + 3) * seg + uk[k]], a);
                        acc[k] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}
// the two layers of a 601-bin head: FC1 with 16-byte loads and groups of 24 rows; FC2 (rows of 601 floats: no alignment) with 4-byte loads
#define MZ_FC1_STREAM 512, true, 24, 1, 8, 3      /* a head alone: one hidden unit per thread */
#define MZ_FC2_STREAM 512, false, 8, 2, 6, 10     /* ... bins t and t + 512 */
#define MZ_FC1_OCTET 512, true, 24, 1, 8, 3       /* octet heads: a wave = one game, lane = unit of the slice (+ the slice length as RS) */
#define MZ_FC2_OCTET 512, false, 16, 3, 6, 10     /* ... lane = bins l, l + 64, l + 128 of the slice */
constexpr int kFcRingFloats = 2 * 3 * 512 * 4 + 8;
inline size_t fcStreamRingFloats(int, int) { return kFcRingFloats; }
__host__ __device__ inline bool fcStream1Fits(int n1, int hidden, int seg) { return n1 % 4 == 0 && hidden % 4 == 0 && seg % 4 == 0 && seg <= 512; }
__host__ __device__ inline bool fcStream2Fits(int hidden, int seg) { return hidden % 4 == 0 && seg <= 1024; }

// conv1x1 (C -> hc channels) + ReLU of a discrete head: xs[C][P] (LDS) -> f[hc * P] (LDS); NT threads, no barrier inside
template <int NT>
__device__ __forceinline__ void discreteConv(const DiscreteParams& d, const float* xs, int C, int P, float* f, int t)
{
    constexpr int K = NT >= 512 ? 2 : 3; // outputs per thread and pass: 612 conv outputs in one pass
    const int n0 = d.hc * P;
    for (int i0 = t; i0 < n0; i0 += NT * K) {
        const float* xk[K];
        const float* wk[K];
        int idx[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            idx[k] = i0 + k * NT < n0 ? i0 + k * NT : n0 - 1; // lanes beyond the end redo the last output and drop it
            const int j = idx[k] / P, p = idx[k] - j * P;
            xk[k] = xs + p;
            wk[k] = d.conv_w + j * C;
        }
        float acc[K];
        dotChainK<32, K, true>(xk, P, wk, 1, C, acc);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (i0 + k * NT < n0) { const float v = acc[k] + d.conv_b[idx[k] / P]; f[idx[k]] = v > 0.0f ? v : 0.0f; }
        }
    }
}

// the same with the conv weights in LDS (wl[hc][C], bl[hc]: copied once per launch by a kernel that runs many simulations): no global-memory
// round trip in front of a 64-step chain
template <int NT>
__device__ __forceinline__ void discreteConvLds(const float* wl, const float* bl, int hc, const float* xs, int C, int P, float* f, int t)
{
    const int n0 = hc * P;
    for (int i0 = t; i0 < n0; i0 += NT * 2) {
        const int i1 = i0 + NT < n0 ? i0 + NT : i0;
        const int j0 = i0 / P, p0 = i0 - j0 * P, j1 = i1 / P, p1 = i1 - j1 * P;
        const float *x0 = xs + p0, *x1 = xs + p1, *w0 = wl + j0 * C, *w1 = wl + j1 * C;
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll 16
        for (int c = 0; c < C; ++c) {
            a0 = __builtin_fmaf(x0[c * P], w0[c], a0);
            a1 = __builtin_fmaf(x1[c * P], w1[c], a1);
        }
        const float v0 = a0 + bl[j0], v1 = a1 + bl[j1];
        f[i0] = v0 > 0.0f ? v0 : 0.0f;
        if (i0 + NT < n0) { f[i1] = v1 > 0.0f ? v1 : 0.0f; }
    }
}

// softmax expectation of the logits lg[size] (LDS; overwritten) -> *out: m = this thread's maximum over the logits it wrote (or -3.4e38f).
// All NT threads; `active` = false: barriers only.  The 601 quotients are independent, only the two index-ordered sums are serial.
template <int NT>
__device__ __forceinline__ void discreteTail(int size, bool active, float m, float* lg, float* red, float* out, int t)
{
    const int lane = t & 63, wave = t >> 6;
    if (active) {
        for (int o = 32; o > 0; o >>= 1) { const float m2 = __shfl_xor(m, o); m = m2 > m ? m2 : m; }
        if (lane == 0) { red[wave] = m; }
    }
    __syncthreads();
    MZ_HPROF(3);
    if (active) {
        m = red[0];
        for (int w = 1; w < NT / 64; ++w) { m = red[w] > m ? red[w] : m; }
        for (int o = t; o < size; o += NT) { lg[o] = mz_expf(lg[o] - m); }
    }
    __syncthreads();
    MZ_HPROF(4);
    if (active && wave == 0) { // index-ordered sum of the exponentials (ref muzero_network.h:157-162)
        const float s = orderedSumWave(lg, size, lane);
        if (lane == 0) { red[8] = s; }
    }
    __syncthreads();
    MZ_HPROF(5);
    if (active) {
        const float s = red[8];
        const int start_value = -size / 2;
        for (int o = t; o < size; o += NT) { lg[o] = (lg[o] / s) * static_cast<float>(start_value + o); } // value * start_value++ (int -> float, exact)
    }
    __syncthreads();
    MZ_HPROF(6);
    if (active && wave == 0) { // accumulate(sum + value * start_value++), in index order
        const float e = orderedSumWave(lg, size, lane);
        if (lane == 0) { *out = e; }
    }
    __syncthreads();
    MZ_HPROF(7);
}

// WIDE (one head per workgroup, sim_cluster.h): one hidden unit / two bins per thread, so that more waves have weight loads in flight
struct NoSideJob {
    __device__ __forceinline__ void operator()() const {}
};

// side / side_mine: a job for the threads with side_mine set (whole waves that have no share of the first FC layer), run beside that layer
template <int NT, bool WIDE = false, class Side = NoSideJob>
__device__ __forceinline__ void discreteHead(const DiscreteParams& d, bool active, const float* xs, int C, int P, float* f, float* h1, float* lg, float* red,
                             float* out, int t, float* ring = nullptr, bool side_mine = false, Side side = Side())
{
    // ring != nullptr (WIDE): FC layers whose shapes fit stream their weights through it (fcStream)
    const bool stream1 = WIDE && ring && NT == 512 && fcStream1Fits(d.hc * P, d.hidden, d.hidden);
    const bool stream2 = WIDE && ring && NT == 512 && fcStream2Fits(d.hidden, d.size);
    if (active) { discreteConv<NT>(d, xs, C, P, f, t); }
    __syncthreads();
    MZ_HPROF(1);
    if (side_mine) {
        side();
    } else if (active) {
        const int n1 = d.hc * P;
        if (WIDE && stream1) {
            if constexpr (WIDE) {
                float acc[1];
                const float* xk[1] = {f};
                const int uk[1] = {t < d.hidden ? t : 0};
                fcStreamSeg<MZ_FC1_STREAM>(xk, uk, t < d.hidden, d.fc1_wT, d.hidden, 0, d.hidden, n1, ring, t, acc);
                if (t < d.hidden) { const float v = acc[0] + d.fc1_b[t]; h1[t] = v > 0.0f ? v : 0.0f; }
            }
        } else if constexpr (WIDE) {
            for (int o = t; o < d.hidden; o += NT) {
                const float v = dotChain<60, true>(f, 1, d.fc1_wT + o, d.hidden, n1) + d.fc1_b[o];
                h1[o] = v > 0.0f ? v : 0.0f;
            }
        } else
        for (int o = 4 * t; o < d.hidden; o += 4 * NT) { // four adjacent hidden units per thread, 16-byte weight loads (dotChain4)
            float acc[4];
            dotChain4<32, true>(f, d.fc1_wT, d.hidden, o, d.hidden, n1, acc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (o + k < d.hidden) { const float v = acc[k] + d.fc1_b[o + k]; h1[o + k] = v > 0.0f ? v : 0.0f; }
            }
        }
    }
    __syncthreads();
    MZ_HPROF(2);
    float m = -3.4e38f;
    if (active && WIDE && stream2) {
        if constexpr (WIDE) {
            float acc[2];
            const float* xk[2] = {h1, h1};
            const int uk[2] = {t < d.size ? t : 0, t + NT < d.size ? t + NT : 0};
            fcStreamSeg<MZ_FC2_STREAM>(xk, uk, t < d.size, d.fc2_wT, d.size, 0, d.size, d.hidden, ring, t, acc);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int o = t + k * NT;
                if (o < d.size) {
                    const float v = acc[k] + d.fc2_b[o];
                    lg[o] = v;
                    m = v > m ? v : m;
                }
            }
        }
    } else if (active && WIDE) {
        for (int o = 2 * t; o < d.size; o += 2 * NT) {
            const int o1 = o + 1 < d.size ? o + 1 : o;
            const float* xk[2] = {h1, h1};
            const float* wk[2] = {d.fc2_wT + o, d.fc2_wT + o1};
            float acc[2];
            dotChainK<30, 2, true>(xk, 1, wk, d.size, d.hidden, acc);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (o + k < d.size) {
                    const float v = acc[k] + d.fc2_b[o + k];
                    lg[o + k] = v;
                    m = v > m ? v : m;
                }
            }
        }
    }
    if (active) {
        for (int o = 4 * t; o < d.size && !WIDE; o += 4 * NT) {
            float acc[4];
            dotChain4<32, true>(h1, d.fc2_wT, d.size, o, d.size, d.hidden, acc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (o + k < d.size) {
                    const float v = acc[k] + d.fc2_b[o + k];
                    lg[o + k] = v;
                    m = v > m ? v : m;
                }
            }
        }
    }
    discreteTail<NT>(d.size, active, m, lg, red, out, t);
}

// invertValue (ref utils/utils.h:102-108) on the device: the reference evaluates the inner expression in double (C `fabs` / `sqrt` on a
// promoted argument), converts to float and squares with powf(x, 2.0f).  sqrt / division in double are IEEE-exact on the device; glibc's
// powf(x, 2.0f) equals the correctly rounded square (float)((double)x * x) — checked on 2e8 random arguments (0 mismatches) and by
// tests/test_gpu_net.py::test_invert_value_on_device against the host function
__device__ __forceinline__ float invertValueDev(float value)
{
    const float epsilon = 0.001;
    const float sign_value = (value > 0.0f ? 1.0f : (value == 0.0f ? 0.0f : -1.0f));
    const double inner = 1 + 4 * epsilon * (__builtin_fabs(static_cast<double>(value)) + 1 + epsilon);
    const float x = static_cast<float>((__builtin_sqrt(inner) - 1) / (2 * epsilon));
    const float sq = static_cast<float>(static_cast<double>(x) * static_cast<double>(x));
    return sign_value * (sq - 1);
}

// The policy head of one sample by ONE wave, without a workgroup barrier: conv1x1 + ReLU, fully connected layer, softmax (every sum the reference's
// sequential chain).  atariHeadsBody runs it on a wave that has no share of the discrete heads' first FC layer, beside that layer.
__device__ __forceinline__ void policyHeadWave(const AtariHeadParams& hp, const float* xs, float* pf, float* lgp, float* __restrict__ policy,
                                               float* __restrict__ logit, int b, int lane)
{
    const int C = hp.C, P = hp.P, A = hp.A, PC = hp.PC;
    for (int i = lane; i < PC * P; i += 64) {
        const int j = i / P, p = i - j * P;
        const float v = dotChain<16, true>(xs + p, P, hp.pconv_w + j * C, 1, C) + hp.pconv_b[j];
        pf[i] = v > 0.0f ? v : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int a = lane; a < A; a += 64) {
        const float v = dotChain<16, true>(pf, 1, hp.pfc_wT + a, A, PC * P) + hp.pfc_b[a];
        lgp[a] = v;
        logit[size_t(b) * A + a] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float m = -3.4e38f;
    for (int a = lane; a < A; a += 64) { m = lgp[a] > m ? lgp[a] : m; }
    for (int o = 32; o > 0; o >>= 1) { const float m2 = __shfl_xor(m, o); m = m2 > m ? m2 : m; }
    for (int a = lane; a < A; a += 64) { lgp[a] = mz_expf(lgp[a] - m); }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float s = 0.0f;
    for (int a = 0; a < A; ++a) { s += lgp[a]; }
    for (int a = lane; a < A; a += 64) { policy[size_t(b) * A + a] = lgp[a] / s; }
}

// The heads of one sample, run by 2 * NTH threads (tid = 0 .. 2 * NTH - 1): x = the trunk's output, either [C][P] in global memory (xg) or
// padded planes in LDS (xlds: channel stride xcs, row stride xpw, 1-pixel border).  `sm` = atariHeadsSmemFloats() floats of LDS scratch.
// invert: value / reward are written in the game's scale (invertValueDev) instead of the transformed scale.
template <int NTH>
__device__ __forceinline__ void atariHeadsBody(const float* __restrict__ xg, const float* __restrict__ xlds, int xcs, int xpw, const AtariHeadParams& hp,
                                               float* __restrict__ policy, float* __restrict__ logit, float* __restrict__ value,
                                               float* __restrict__ reward, float* __restrict__ hd, int do_reward, int invert, int b, int tid,
                                               float* __restrict__ sm)
{
    constexpr int NT2 = 2 * NTH;
    const int C = hp.C, P = hp.P, A = hp.A, PC = hp.PC;
    const int hcmax = hp.value.hc > hp.reward.hc ? hp.value.hc : hp.reward.hc;
    const int hidmax = hp.value.hidden > hp.reward.hidden ? hp.value.hidden : hp.reward.hidden;
    const int sizemax = hp.value.size > hp.reward.size ? hp.value.size : hp.reward.size;
    const int lane = tid & 63, wave = tid >> 6, half = tid / NTH, t = tid - half * NTH;
    const int per_half = hcmax * P + hidmax + sizemax + 16;
    float* xr = sm;                     // [C*P] the trunk's output (the reward head reads the UNscaled hidden state)
    float* xs = xr + C * P;             // [C*P] the rescaled hidden state (policy and value heads, and the slab)
    float* pf = xs + C * P;             // [PC*P]
    float* lgp = pf + PC * P;           // [A]
    float* redp = lgp + A;              // [32]
    float* hb = redp + 32 + half * per_half; // this half's head scratch
    float* f = hb;                      // [hcmax*P]
    float* h1 = f + hcmax * P;          // [hidmax]
    float* lg = h1 + hidmax;            // [sizemax]
    float* red = lg + sizemax;          // [16]
    if (xlds) {
        const int Wb = xpw - 2;
        for (int i = tid; i < C * P; i += NT2) {
            const int c = i / P, p = i - c * P;
            xr[i] = xlds[c * xcs + (p / Wb + 1) * xpw + p % Wb + 1];
        }
    } else {
        for (int i = tid; i < C * P; i += NT2) { xr[i] = xg[i]; }
    }
    __syncthreads();
    // scale_hidden_state (ref muzero_atari_network.py:189-198)
    {
        float mn = 3.4e38f, mx = -3.4e38f;
        for (int i = tid; i < C * P; i += NT2) { const float v = xr[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(mn, o), x2 = __shfl_xor(mx, o);
            mn = m2 < mn ? m2 : mn;
            mx = x2 > mx ? x2 : mx;
        }
        if (lane == 0) { redp[wave] = mn; redp[16 + wave] = mx; }
        __syncthreads();
        mn = redp[0]; mx = redp[16];
        for (int w = 1; w < NT2 / 64; ++w) { mn = redp[w] < mn ? redp[w] : mn; mx = redp[16 + w] > mx ? redp[16 + w] : mx; }
        float scale = mx - mn;
        if (scale < 1e-5f) { scale += 1e-5f; }
        for (int i = tid; i < C * P; i += NT2) {
            const float v = (xr[i] - mn) / scale;
            xs[i] = v;
            hd[i] = v;
        }
        __syncthreads();
    }
    // half 0: reward head on the unscaled state (ref muzero_atari_network.py: dynamics -> reward before the rescale);
    // half 1: value head on the rescaled state
    MZ_HPROF(0);
    float* out = half == 0 ? reward + b : value + b;
    // the policy head rides on the last wave of the value half while the first FC layers run (four hidden units per thread: the layer occupies the
    // first hidmax / 4 threads of each half); 4.8 us of 40 per leaf on BASELINE configs[4] when it followed the discrete heads on all threads
    const bool policy_beside = 4 * (NTH - 64) >= hidmax;
    auto policy_job = [&]() { policyHeadWave(hp, xs, pf, lgp, policy, logit, b, lane); };
    discreteHead<NTH, false, decltype(policy_job)>(half == 0 ? hp.reward : hp.value, half == 0 ? do_reward != 0 : true, half == 0 ? xr : xs, C, P, f, h1, lg, red,
                                                        out, t, nullptr, policy_beside && tid >= NT2 - 64, policy_job);
    if (invert && t == 0 && (half == 1 || do_reward)) { *out = invertValueDev(*out); }
    if (policy_beside) { return; }
    // policy head (all threads; its barriers come after the discrete heads')
    for (int i = tid; i < PC * P; i += NT2) {
        const int j = i / P, p = i - j * P;
        const float v = dotChain<16, true>(xs + p, P, hp.pconv_w + j * C, 1, C) + hp.pconv_b[j];
        pf[i] = v > 0.0f ? v : 0.0f;
    }
    __syncthreads();
    MZ_HPROF(8);
    for (int a = tid; a < A; a += NT2) {
        const float v = dotChain<16, true>(pf, 1, hp.pfc_wT + a, A, PC * P) + hp.pfc_b[a];
        lgp[a] = v;
        logit[size_t(b) * A + a] = v;
    }
    __syncthreads();
    MZ_HPROF(9);
    if (wave == 0) {
        float m = -3.4e38f;
        for (int a = lane; a < A; a += 64) { m = lgp[a] > m ? lgp[a] : m; }
        for (int o = 32; o > 0; o >>= 1) { const float m2 = __shfl_xor(m, o); m = m2 > m ? m2 : m; }
        for (int a = lane; a < A; a += 64) { lgp[a] = mz_expf(lgp[a] - m); }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float s = 0.0f;
        for (int a = 0; a < A; ++a) { s += lgp[a]; }
        for (int a = lane; a < A; a += 64) { policy[size_t(b) * A + a] = lgp[a] / s; }
    }
    MZ_HPROF(10);
}

inline size_t atariHeadsSmemFloats(const AtariHeadParams& hp)
{
    const int hcmax = hp.value.hc > hp.reward.hc ? hp.value.hc : hp.reward.hc;
    const int hidmax = hp.value.hidden > hp.reward.hidden ? hp.value.hidden : hp.reward.hidden;
    const int sizemax = hp.value.size > hp.reward.size ? hp.value.size : hp.reward.size;
    return size_t(2) * hp.C * hp.P + size_t(hp.PC) * hp.P + hp.A + 32 + size_t(2) * (size_t(hcmax) * hp.P + hidmax + sizemax + 16);
}

} // namespace mz
