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
// s = ((x[0] + x[1]) + x[2]) + ... in index order, by ONE wave: lane l holds elements [l * VPL, (l + 1) * VPL) in registers and the running
// sum is handed from lane to lane (v_readlane).  The same n - 1 dependent adds as a scalar loop, but without an LDS round trip per
// element (one lane reading x[i] and adding, 601 times, cost 33 us per sum: the 601-bin heads have four such sums).
template <int VPL>
__device__ __forceinline__ float orderedSumWaveT(const float* x, int n, int vpl, int lane)
{
    float v[VPL]; // slots beyond the lane's elements hold +0: adding +0 never changes a sum that is not -0, and these sums never are
#pragma unroll
    for (int k = 0; k < VPL; ++k) { const int i = lane * vpl + k; v[k] = (k < vpl && i < n) ? x[i] : 0.0f; }
    float acc = 0.0f;
    const int lanes = (n + vpl - 1) / vpl;
    for (int l = 0; l < lanes; ++l) { // straight-line body: VPL dependent adds in every lane, lane l's result is the one that counts
        float a = acc;
#pragma unroll
        for (int k = 0; k < VPL; ++k) { a = a + v[k]; }
        acc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), l));
    }
    return acc;
}
__device__ __forceinline__ float orderedSumWave(const float* x, int n, int lane)
{
    const int vpl = (n + 63) / 64;
    if (vpl <= 4) { return orderedSumWaveT<4>(x, n, vpl, lane); }
    if (vpl <= 10) { return orderedSumWaveT<10>(x, n, vpl, lane); } // 601 bins
    if (vpl <= 16) { return orderedSumWaveT<16>(x, n, vpl, lane); }
    float s = 0.0f; // not reached by the supported head sizes: plain loop
    for (int i = 0; i < n; ++i) { s += x[i]; }
    return s;
}

// One fully connected layer with the weights STREAMED through LDS (sim_cluster.h: a head alone on its CU).  y[o] = the ordered f32 chain over
// i < n of x[i] * W[i][o] (weights wT[n][ws], 16-byte aligned), o = t and, K == 2, t + NT.  The chain length is fixed, so its time is
// (rows / rows in flight) x memory latency — and the weights of the 601-bin heads do not stay in the 4-MB L2 next to the tower's
// (measured: 1.3 us per batch of 60 rows, FC1 14 us).  Here ALL NT threads fetch chunk c + D - 1 (R rows, 16-byte loads, D - 1 chunks in flight in
// registers = 170 KB) while chunk c is consumed from an LDS ring of two chunks, so the layer runs at the CU's L2-port bandwidth whatever the
// hit rate.  Needs n % 4 == 0, (R * ws) % 4 == 0, LPT * NT * 4 >= R * ws; ring = 2 * R * ws floats (16-byte aligned), x 16-byte aligned.
template <int NT, int R, int D, int LPT, int K, int NCH>
__device__ __forceinline__ void fcStream(const float* __restrict__ x, const float* __restrict__ W, int ws, int n, int nout, float* __restrict__ ring, int t,
                                         float (&acc)[K])
{
    typedef float vf4 __attribute__((ext_vector_type(4)));
    static_assert(R % 4 == 0, "four rows per step");
    constexpr int SLOT = LPT * NT * 4; // floats per ring slot: every thread writes all its LPT vectors, whatever the chunk size
    const int chunk4 = R * ws / 4, nchunks = (n + R - 1) / R, total4 = n * (ws / 4) + n * (ws % 4) / 4; // = n * ws / 4
    // the weights are in global memory: say so, or the loads are FLAT loads, which also count on lgkmcnt — every wait for an LDS read would then
    // wait for all weight loads in flight
    typedef __attribute__((address_space(1))) const vf4 GV4;
    GV4* W4 = (GV4*)(W);
    vf4 buf[D][LPT];
    int o[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { acc[k] = 0.0f; o[k] = t + k * NT < nout ? t + k * NT : 0; }
    // No branch around a global load or around an iteration (chunks beyond the layer fetch element 0 and compute nothing): the compiler counts the
    // loads in flight per PATH and waits for the oldest chunk with s_waitcnt vmcnt(N), N = the loads it can prove to be younger — with
    // conditional iterations N shrinks towards 0 and every chunk pays the full memory latency (measured: 1 us per chunk).  Straight-line for
    // the same reason: inside a loop the prefetch would be loop-carried and waited for with vmcnt(0).
    auto issue = [&](vf4 (&b)[LPT], int c) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int k4 = t + j * NT, g4 = c * chunk4 + k4;
            const bool ok = c < nchunks && k4 < chunk4 && g4 < total4;
            b[j] = W4[ok ? g4 : 0];
        }
    };
    auto drain = [&](const vf4 (&b)[LPT], int c) {
        vf4* dst = reinterpret_cast<vf4*>(ring + size_t(c & 1) * SLOT);
#pragma unroll
        for (int j = 0; j < LPT; ++j) { dst[t + j * NT] = b[j]; }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) { issue(buf[d], d); }
    drain(buf[0], 0);
    __syncthreads();
    const bool mine = t < nout;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        issue(buf[c % D], c + D);
        drain(buf[(c + 1) % D], c + 1);
        const int rows = n - c * R < R ? n - c * R : R;
        if (mine && rows > 0) {
            const float* rs = ring + size_t(c & 1) * SLOT;
            auto step4 = [&](int r) {
                const vf4 xv = *reinterpret_cast<const vf4*>(x + c * R + r);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float a = acc[k];
                    a = __builtin_fmaf(xv.x, rs[(r + 0) * ws + o[k]], a);
                    a = __builtin_fmaf(xv.y, rs[(r + 1) * ws + o[k]], a);
                    a = __builtin_fmaf(xv.z, rs[(r + 2) * ws + o[k]], a);
                    a = __builtin_fmaf(xv.w, rs[(r + 3) * ws + o[k]], a);
                    acc[k] = a;
                }
            };
            if (rows == R) { // all LDS reads of the chunk first, then the dependent fmas (left alone the scheduler puts every read right in
                             // front of its fma: 24 LDS round trips in a row, 0.9 us per chunk)
                vf4 xv[R / 4];
                float wv[K][R];
#pragma unroll
                for (int r = 0; r < R; r += 4) { xv[r / 4] = *reinterpret_cast<const vf4*>(x + c * R + r); }
#pragma unroll
                for (int k = 0; k < K; ++k) {
#pragma unroll
                    for (int r = 0; r < R; ++r) { wv[k][r] = rs[r * ws + o[k]]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float a = acc[k];
#pragma unroll
                    for (int r = 0; r < R; r += 4) {
                        a = __builtin_fmaf(xv[r / 4].x, wv[k][r + 0], a);
                        a = __builtin_fmaf(xv[r / 4].y, wv[k][r + 1], a);
                        a = __builtin_fmaf(xv[r / 4].z, wv[k][r + 2], a);
                        a = __builtin_fmaf(xv[r / 4].w, wv[k][r + 3], a);
                    }
                    acc[k] = a;
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                for (int r = 0; r < rows; r += 4) { step4(r); }
            }
        }
        __syncthreads();
    }
}
constexpr int kFcStreamR1 = 24, kFcStreamR2 = 8; // rows per chunk of FC1 / FC2
constexpr int kFcStreamN1 = 26, kFcStreamN2 = 32; // chunks the straight-line code covers (FC1: <= 624 inputs, FC2: <= 256)
inline size_t fcStreamRingFloats(int, int) { return size_t(2) * 3 * 512 * 4 + 8; } // two slots of LPT * NT 16-byte vectors

// WIDE (one head per workgroup, sim_cluster.h): one hidden unit / two bins per thread, so that more waves have weight loads in flight
template <int NT, bool WIDE = false>
__device__ __forceinline__ void discreteHead(const DiscreteParams& d, bool active, const float* xs, int C, int P, float* f, float* h1, float* lg, float* red,
                             float* out, int t, float* ring = nullptr)
{
    // ring != nullptr (WIDE): FC layers whose shapes fit stream their weights through it (fcStream)
    const bool stream1 = WIDE && ring && (d.hc * P) % 4 == 0 && d.hidden % 4 == 0 && d.hidden <= NT && kFcStreamR1 * d.hidden <= 3 * NT * 4 && d.hc * P <= kFcStreamN1 * kFcStreamR1;
    const bool stream2 = WIDE && ring && d.hidden % 4 == 0 && d.size <= 2 * NT && kFcStreamR2 * d.size <= 3 * NT * 4 && d.hidden <= kFcStreamN2 * kFcStreamR2;
    const int lane = t & 63, wave = t >> 6;
    constexpr int K = NT >= 512 ? 2 : 3; // outputs per thread and pass: 601 bins / 612 conv outputs in one pass of the half
    if (active) {
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
            dotChainK<32, K>(xk, P, wk, 1, C, acc);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (i0 + k * NT < n0) { const float v = acc[k] + d.conv_b[idx[k] / P]; f[idx[k]] = v > 0.0f ? v : 0.0f; }
            }
        }
    }
    __syncthreads();
    MZ_HPROF(1);
    if (active) {
        const int n1 = d.hc * P;
        if (WIDE && stream1) {
            if constexpr (WIDE) {
                float acc[1];
                fcStream<NT, kFcStreamR1, 8, 3, 1, kFcStreamN1>(f, d.fc1_wT, d.hidden, n1, d.hidden, ring, t, acc);
                if (t < d.hidden) { const float v = acc[0] + d.fc1_b[t]; h1[t] = v > 0.0f ? v : 0.0f; }
            }
        } else if constexpr (WIDE) {
            for (int o = t; o < d.hidden; o += NT) {
                const float v = dotChain<60>(f, 1, d.fc1_wT + o, d.hidden, n1) + d.fc1_b[o];
                h1[o] = v > 0.0f ? v : 0.0f;
            }
        } else
        for (int o = 4 * t; o < d.hidden; o += 4 * NT) { // four adjacent hidden units per thread, 16-byte weight loads (dotChain4)
            float acc[4];
            dotChain4<32>(f, d.fc1_wT, d.hidden, o, d.hidden, n1, acc);
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
            fcStream<NT, kFcStreamR2, 8, 3, 2, kFcStreamN2>(h1, d.fc2_wT, d.size, d.hidden, d.size, ring, t, acc);
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
            dotChainK<30, 2>(xk, 1, wk, d.size, d.hidden, acc);
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
            dotChain4<32>(h1, d.fc2_wT, d.size, o, d.size, d.hidden, acc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (o + k < d.size) {
                    const float v = acc[k] + d.fc2_b[o + k];
                    lg[o + k] = v;
                    m = v > m ? v : m;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) { const float m2 = __shfl_xor(m, o); m = m2 > m ? m2 : m; }
        if (lane == 0) { red[wave] = m; }
    }
    __syncthreads();
    MZ_HPROF(3);
    if (active) {
        m = red[0];
        for (int w = 1; w < NT / 64; ++w) { m = red[w] > m ? red[w] : m; }
        for (int o = t; o < d.size; o += NT) { lg[o] = mz_expf(lg[o] - m); }
    }
    __syncthreads();
    MZ_HPROF(4);
    if (active && wave == 0) { // index-ordered sum of the exponentials (ref muzero_network.h:157-162)
        const float s = orderedSumWave(lg, d.size, lane);
        if (lane == 0) { red[8] = s; }
    }
    __syncthreads();
    MZ_HPROF(5);
    if (active) {
        const float s = red[8];
        const int start_value = -d.size / 2;
        for (int o = t; o < d.size; o += NT) { lg[o] = (lg[o] / s) * static_cast<float>(start_value + o); } // value * start_value++ (int -> float, exact)
    }
    __syncthreads();
    MZ_HPROF(6);
    if (active && wave == 0) { // accumulate(sum + value * start_value++), in index order
        const float e = orderedSumWave(lg, d.size, lane);
        if (lane == 0) { *out = e; }
    }
    __syncthreads();
    MZ_HPROF(7);
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
    discreteHead<NTH>(half == 0 ? hp.reward : hp.value, half == 0 ? do_reward != 0 : true, half == 0 ? xr : xs, C, P, f, h1, lg, red, out, t);
    if (invert && t == 0 && (half == 1 || do_reward)) { *out = invertValueDev(*out); }
    // policy head (all threads; its barriers come after the discrete heads')
    for (int i = tid; i < PC * P; i += NT2) {
        const int j = i / P, p = i - j * P;
        const float v = dotChain<16>(xs + p, P, hp.pconv_w + j * C, 1, C) + hp.pconv_b[j];
        pf[i] = v > 0.0f ? v : 0.0f;
    }
    __syncthreads();
    for (int a = tid; a < A; a += NT2) {
        const float v = dotChain<16>(pf, 1, hp.pfc_wT + a, A, PC * P) + hp.pfc_b[a];
        lgp[a] = v;
        logit[size_t(b) * A + a] = v;
    }
    __syncthreads();
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
}

inline size_t atariHeadsSmemFloats(const AtariHeadParams& hp)
{
    const int hcmax = hp.value.hc > hp.reward.hc ? hp.value.hc : hp.reward.hc;
    const int hidmax = hp.value.hidden > hp.reward.hidden ? hp.value.hidden : hp.reward.hidden;
    const int sizemax = hp.value.size > hp.reward.size ? hp.value.size : hp.reward.size;
    return size_t(2) * hp.C * hp.P + size_t(hp.PC) * hp.P + hp.A + 32 + size_t(2) * (size_t(hcmax) * hp.P + hidmax + sizemax + 16);
}

} // namespace mz
