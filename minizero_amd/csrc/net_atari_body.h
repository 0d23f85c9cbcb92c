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

template <int NT>
__device__ __forceinline__ void discreteHead(const DiscreteParams& d, bool active, const float* xs, int C, int P, float* f, float* h1, float* lg, float* red,
                             float* out, int t)
{
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
    if (active) {
        for (int o = 4 * t; o < d.size; o += 4 * NT) {
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
