// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  CPU restatement of the network math of
// ref network/py/{network_unit,alphazero_network,muzero_network}.py in eval mode.
//
// Pinned by tests/golden/nn_*.npz (outputs of the reference's own Python modules, f32 CPU) at
// |diff| <= 1e-5.  The arithmetic ORDER below is a specification shared with the HIP kernels so the
// GPU path can be compared bit-for-bit, not only within the north_star's 1e-3:
//   * BatchNorm (eval, eps 1e-5) is folded on the host:  s = gamma / sqrtf(var + eps);
//     w' = w * s;  b' = (b - mean) * s + beta                      (f32, no contraction)
//   * conv3x3 (pad 1): acc = 0; for tap t = ky*3+kx (0..8), for c = 0..C_in-1:
//         acc = fmaf(x[c][y+ky-1][x+kx-1] (0 outside), w'[oc][c][ky][kx], acc)
//     y = acc + b'[oc]; (+ skip); relu.           (== an MFMA f32 k-ordered fma chain)
//   * conv1x1 / linear: acc = 0; for i ascending: acc = fmaf(in[i], w[out][i], acc); + bias
//   * softmax: m = max; e_i = mz_expf(l_i - m); s = sum in index order; p_i = e_i / s
//   * tanh: mz_tanhf.   scale_hidden_state: min/max exact, (h - min) / scale
#include "oracle.h"
#include <algorithm>
#include <cassert>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <immintrin.h>
#include <thread>

namespace mzo {

// persistent worker pool for the per-sample loops of the forward pass (cpu_baseline: no thread creation per call)
namespace {
class NNPool {
public:
    static NNPool& get() { static NNPool* p = new NNPool(); return *p; } // leaked on purpose: detached workers must never see it destroyed
    void run(int n, const std::function<void(int)>& f)
    {
        if (n <= 1 || nt_ <= 1) { for (int i = 0; i < n; ++i) { f(i); } return; }
        std::unique_lock<std::mutex> l(mu_);
        fn_ = &f; n_ = n; next_ = 0; done_ = 0; ++epoch_;
        cv_.notify_all();
        l.unlock();
        work();
        l.lock();
        dcv_.wait(l, [&] { return done_ == n_; });
    }
private:
    NNPool()
    {
        // threads = CPUs this container may actually use (cgroup quota), else 256 runnable threads get throttled on a 16-CPU quota
        nt_ = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
        if (const char* e = getenv("MZO_THREADS")) { nt_ = std::max(1, atoi(e)); }
        else if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            double period = 100000;
            if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) { nt_ = std::max(1, std::min(nt_, static_cast<int>(atof(q) / period))); }
            fclose(f);
        }
        for (int t = 1; t < nt_; ++t) { std::thread([this] { uint64_t seen = 0; for (;;) { { std::unique_lock<std::mutex> l(mu_); cv_.wait(l, [&] { return epoch_ != seen; }); seen = epoch_; } work(); } }).detach(); }
    }
    void work()
    {
        for (;;) {
            int i;
            { std::lock_guard<std::mutex> l(mu_); if (next_ >= n_) { return; } i = next_++; }
            (*fn_)(i);
            { std::lock_guard<std::mutex> l(mu_); if (++done_ == n_) { dcv_.notify_all(); } }
        }
    }
    int nt_ = 1, n_ = 0, next_ = 0, done_ = 0;
    uint64_t epoch_ = 0;
    const std::function<void(int)>* fn_ = nullptr;
    std::mutex mu_;
    std::condition_variable cv_, dcv_;
};
} // namespace

float mz_expf(float x)
{
    if (x < -87.0f) { return 0.0f; }
    if (x > 88.0f) { x = 88.0f; }
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int ni = static_cast<int>(n);
    uint32_t bits = static_cast<uint32_t>(ni + 127) << 23;
    float scale;
    memcpy(&scale, &bits, 4);
    return y * scale;
}

float mz_tanhf(float x)
{
    float ax = fabsf(x);
    if (ax > 10.0f) { return copysignf(1.0f, x); }
    float e = mz_expf(-2.0f * ax);
    float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}

// ---- raw parameter manifest (state_dict order of the reference modules, num_batches_tracked skipped) ----
enum Kind { W, B, BN_G, BN_B, BN_M, BN_V };
struct TensorSpec { size_t n; int fan_in; Kind kind; };

static void specConvBN(std::vector<TensorSpec>& m, int cin, int cout, int k)
{
    m.push_back({size_t(cout) * cin * k * k, cin * k * k, W});
    m.push_back({size_t(cout), cin * k * k, B});
    m.push_back({size_t(cout), 0, BN_G});
    m.push_back({size_t(cout), 0, BN_B});
    m.push_back({size_t(cout), 0, BN_M});
    m.push_back({size_t(cout), 0, BN_V});
}
static void specLinear(std::vector<TensorSpec>& m, int in, int out)
{
    m.push_back({size_t(out) * in, in, W});
    m.push_back({size_t(out), in, B});
}
static int policyChannels(const NetDesc& d) { int hw = d.hidden_channel_height * d.hidden_channel_width; return (d.action_size + hw - 1) / hw; }

static std::vector<TensorSpec> manifest(const NetDesc& d)
{
    std::vector<TensorSpec> m;
    const int C = d.num_hidden_channels, hw = d.hidden_channel_height * d.hidden_channel_width;
    auto trunk = [&](int cin) { // ref alphazero_network.py:32-34 / muzero_network.py:10-12,27-29
        specConvBN(m, cin, C, 3);
        for (int b = 0; b < d.num_blocks; ++b) { specConvBN(m, C, C, 3); specConvBN(m, C, C, 3); } // ref network_unit.py:9-12
    };
    auto heads = [&]() { // ref network_unit.py:26-64 (PolicyNetwork, ValueNetwork)
        int pc = policyChannels(d);
        specConvBN(m, C, pc, 1);
        specLinear(m, pc * hw, d.action_size);
        specConvBN(m, C, 1, 1);
        specLinear(m, hw, d.num_value_hidden_channels);
        specLinear(m, d.num_value_hidden_channels, 1);
    };
    if (d.type == 0) {
        trunk(d.num_input_channels);
        heads();
    } else if (d.type == 1) { // muzero: representation, dynamics, prediction (ref muzero_network.py:79-81)
        trunk(d.num_input_channels);
        trunk(C + d.num_action_feature_channels);
        heads();
    } else { // muzero_atari (ref muzero_atari_network.py:7-70,116-118)
        auto rb = [&](int ch) { specConvBN(m, ch, ch, 3); specConvBN(m, ch, ch, 3); };
        auto discrete = [&](int hidden, int size) { // DiscreteValueNetwork (ref network_unit.py:67-87)
            int hc = (size + hw - 1) / hw;
            specConvBN(m, C, hc, 1);
            specLinear(m, hw * hc, hidden);
            specLinear(m, hidden, size);
        };
        // representation: conv1 s2, bn1, RB(C/2), conv2 s2, bn2, RB(C), [pool], RB(C), [pool], num_blocks x RB(C)
        specConvBN(m, d.num_input_channels, C / 2, 3);
        rb(C / 2);
        specConvBN(m, C / 2, C, 3);
        rb(C);
        rb(C);
        for (int b = 0; b < d.num_blocks; ++b) { rb(C); }
        // dynamics: conv, bn, blocks, reward_network
        specConvBN(m, C + d.num_action_feature_channels, C, 3);
        for (int b = 0; b < d.num_blocks; ++b) { rb(C); }
        discrete(C, d.discrete_value_size);
        // prediction: policy, value
        int pc = policyChannels(d);
        specConvBN(m, C, pc, 1);
        specLinear(m, pc * hw, d.action_size);
        discrete(d.num_value_hidden_channels, d.discrete_value_size);
    }
    return m;
}

size_t Net::rawParamCount(const NetDesc& d)
{
    size_t n = 0;
    for (auto& t : manifest(d)) { n += t.n; }
    return n;
}

static inline uint64_t mix64(uint64_t z) // splitmix64 finaliser, counter based
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void Net::generateRaw(const NetDesc& d, uint64_t seed, float* out)
{
    size_t idx = 0;
    for (auto& t : manifest(d)) {
        float lo, hi;
        switch (t.kind) {
            case W:
            case B: { float bound = 1.0f / sqrtf(static_cast<float>(t.fan_in)); lo = -bound; hi = bound; break; }
            case BN_G: lo = 0.5f; hi = 1.5f; break;
            case BN_V: lo = 0.5f; hi = 1.5f; break;
            default: lo = -0.1f; hi = 0.1f; break;
        }
        for (size_t i = 0; i < t.n; ++i, ++idx) {
            uint64_t z = mix64(seed + (idx + 1) * 0x9E3779B97F4A7C15ULL);
            float u = static_cast<float>(z >> 40) * 5.9604644775390625e-08f; // 2^-24, exact in f32
            out[idx] = lo + (hi - lo) * u;
        }
    }
}

// ---- folded layers ----
struct Conv {
    int cin, cout, k;
    std::vector<float> w, b;  // w[oc][c][ky][kx] folded, b folded
    std::vector<float> wk;    // 3x3 only: the same weights as wk[t][c][oc] (oc padded with zeros to a multiple of 32) for convChains(), built once (finishConv)
    int cout_pad = 0;
};
struct Linear { int in, out; std::vector<float> w, b; };

static void finishConv(Conv& c);
static Conv takeConvBN(const float*& p, int cin, int cout, int k)
{
    Conv c{cin, cout, k, {}, {}};
    size_t nw = size_t(cout) * cin * k * k;
    const float *w = p, *b = p + nw, *g = b + cout, *be = g + cout, *mu = be + cout, *var = mu + cout;
    p = var + cout;
    c.w.resize(nw);
    c.b.resize(cout);
    for (int oc = 0; oc < cout; ++oc) {
        float s = g[oc] / sqrtf(var[oc] + 1e-5f);
        for (int i = 0; i < cin * k * k; ++i) { c.w[size_t(oc) * cin * k * k + i] = w[size_t(oc) * cin * k * k + i] * s; }
        float t = (b[oc] - mu[oc]) * s;
        c.b[oc] = t + be[oc];
    }
    finishConv(c);
    return c;
}
static Linear takeLinear(const float*& p, int in, int out)
{
    Linear l{in, out, {}, {}};
    l.w.assign(p, p + size_t(in) * out);
    p += size_t(in) * out;
    l.b.assign(p, p + out);
    p += out;
    return l;
}

// The (tap, channel)-ordered fmaf chain of every (pixel, output channel) of a 3x3 convolution, pad 1, stride S — the arithmetic of the header, one IEEE fma per
// step, nothing reassociated: the chains of 4 pixels x 16 output channels advance side by side in registers (8 lanes of a ymm = 8 output channels; vfmadd is
// the same correctly rounded fma as fmaf), which only changes how many chains are in flight, not one bit of any of them.  A tap outside the plane multiplies
// by 0.0f like the scalar loop did.  (The scalar loop kept its accumulators in memory: 4 GFLOP/s per thread; this one runs at the FMA ports' rate.)
static void finishConv(Conv& c)
{
    if (c.k != 3) { return; }
    c.cout_pad = (c.cout + 31) / 32 * 32; // (blocks of 16 output channels on AVX2, of 32 on AVX-512)
    c.wk.assign(size_t(9) * c.cin * c.cout_pad, 0.0f);
    for (int oc = 0; oc < c.cout; ++oc)
        for (int ch = 0; ch < c.cin; ++ch)
            for (int t = 0; t < 9; ++t) { c.wk[(size_t(t) * c.cin + ch) * c.cout_pad + oc] = c.w[(size_t(oc) * c.cin + ch) * 9 + t]; }
}

template <int NP>
static inline void convChains(const Conv& cv, int H, int Wd, int stride, int Wo, const float* in, const float* skip, float* out, int p0, int Po)
{
    static const float kZero = 0.0f;
    const int cin = cv.cin, cout = cv.cout, cp = cv.cout_pad, Pi = H * Wd;
    const float* base[9][NP];
    size_t step[9][NP];
    for (int j = 0; j < NP; ++j) {
        const int p = p0 + j, y = p / Wo, x = p - y * Wo;
        for (int t = 0; t < 9; ++t) {
            const int yy = y * stride + t / 3 - 1, xx = x * stride + t % 3 - 1;
            const bool inside = (yy >= 0 && yy < H && xx >= 0 && xx < Wd);
            base[t][j] = inside ? in + yy * Wd + xx : &kZero;
            step[t][j] = inside ? size_t(Pi) : 0;
        }
    }
    for (int o0 = 0; o0 < cp; o0 += 16) {
        __m256 acc[NP][2];
        for (int j = 0; j < NP; ++j) { acc[j][0] = _mm256_setzero_ps(); acc[j][1] = _mm256_setzero_ps(); }
        for (int t = 0; t < 9; ++t) {
            const float* wr = &cv.wk[size_t(t) * cin * cp + o0];
            const float* bp[NP];
            size_t st[NP];
            for (int j = 0; j < NP; ++j) { bp[j] = base[t][j]; st[j] = step[t][j]; }
            for (int c = 0; c < cin; ++c, wr += cp) {
                const __m256 w0 = _mm256_loadu_ps(wr), w1 = _mm256_loadu_ps(wr + 8);
                for (int j = 0; j < NP; ++j) {
                    const __m256 xv = _mm256_broadcast_ss(bp[j]);
                    bp[j] += st[j];
                    acc[j][0] = _mm256_fmadd_ps(xv, w0, acc[j][0]);
                    acc[j][1] = _mm256_fmadd_ps(xv, w1, acc[j][1]);
                }
            }
        }
        for (int j = 0; j < NP; ++j) {
            alignas(32) float a[16];
            _mm256_store_ps(a, acc[j][0]);
            _mm256_store_ps(a + 8, acc[j][1]);
            const int p = p0 + j;
            for (int i = 0; i < 16 && o0 + i < cout; ++i) {
                const int oc = o0 + i;
                float v = a[i] + cv.b[oc];
                if (skip) { v = v + skip[size_t(oc) * Po + p]; }
                out[size_t(oc) * Po + p] = v > 0.0f ? v : 0.0f;
            }
        }
    }
}

// The same chains on AVX-512 where the CPU has it (run-time dispatch; MZO_NO_AVX512=1 keeps the AVX2 path): 8 pixels x 32 output channels = 16 accumulators of 16
// lanes.  vfmadd231ps on a zmm lane is the same correctly rounded fma: the outputs are the same bits on either path (convSelfTest runs both against the scalar chain).
template <int NP>
__attribute__((target("avx512f"))) static inline void convChains512(const Conv& cv, int H, int Wd, int stride, int Wo, const float* in, const float* skip, float* out, int p0, int Po)
{
    static const float kZero = 0.0f;
    const int cin = cv.cin, cout = cv.cout, cp = cv.cout_pad, Pi = H * Wd;
    const float* base[9][NP];
    size_t step[9][NP];
    for (int j = 0; j < NP; ++j) {
        const int p = p0 + j, y = p / Wo, x = p - y * Wo;
        for (int t = 0; t < 9; ++t) {
            const int yy = y * stride + t / 3 - 1, xx = x * stride + t % 3 - 1;
            const bool inside = (yy >= 0 && yy < H && xx >= 0 && xx < Wd);
            base[t][j] = inside ? in + yy * Wd + xx : &kZero;
            step[t][j] = inside ? size_t(Pi) : 0;
        }
    }
    for (int o0 = 0; o0 < cp; o0 += 32) {
        __m512 acc[NP][2];
        for (int j = 0; j < NP; ++j) { acc[j][0] = _mm512_setzero_ps(); acc[j][1] = _mm512_setzero_ps(); }
        for (int t = 0; t < 9; ++t) {
            const float* wr = &cv.wk[size_t(t) * cin * cp + o0];
            const float* bp[NP];
            size_t st[NP];
            for (int j = 0; j < NP; ++j) { bp[j] = base[t][j]; st[j] = step[t][j]; }
            for (int c = 0; c < cin; ++c, wr += cp) {
                const __m512 w0 = _mm512_loadu_ps(wr), w1 = _mm512_loadu_ps(wr + 16);
                for (int j = 0; j < NP; ++j) {
                    const __m512 xv = _mm512_set1_ps(*bp[j]);
                    bp[j] += st[j];
                    acc[j][0] = _mm512_fmadd_ps(xv, w0, acc[j][0]);
                    acc[j][1] = _mm512_fmadd_ps(xv, w1, acc[j][1]);
                }
            }
        }
        for (int j = 0; j < NP; ++j) {
            alignas(64) float a[32];
            _mm512_store_ps(a, acc[j][0]);
            _mm512_store_ps(a + 16, acc[j][1]);
            const int p = p0 + j;
            for (int i = 0; i < 32 && o0 + i < cout; ++i) {
                const int oc = o0 + i;
                float v = a[i] + cv.b[oc];
                if (skip) { v = v + skip[size_t(oc) * Po + p]; }
                out[size_t(oc) * Po + p] = v > 0.0f ? v : 0.0f;
            }
        }
    }
}
__attribute__((target("avx512f"))) static void conv3x3s512(const Conv& cv, int H, int Wd, int stride, const float* in, const float* skip, float* out)
{
    const int Ho = (H - 1) / stride + 1, Wo = (Wd - 1) / stride + 1, Po = Ho * Wo;
    int p = 0;
    for (; p + 8 <= Po; p += 8) { convChains512<8>(cv, H, Wd, stride, Wo, in, skip, out, p, Po); }
    for (; p + 2 <= Po; p += 2) { convChains512<2>(cv, H, Wd, stride, Wo, in, skip, out, p, Po); }
    for (; p < Po; ++p) { convChains512<1>(cv, H, Wd, stride, Wo, in, skip, out, p, Po); }
}
static bool useAvx512()
{
    static const bool yes = __builtin_cpu_supports("avx512f") && getenv("MZO_NO_AVX512") == nullptr;
    return yes;
}

// conv3x3 pad 1 with stride, optional skip, relu.  in[cin][H][W] -> out[cout][Ho][Wo]
static void conv3x3sAvx2(const Conv& cv, int H, int Wd, int stride, const float* in, const float* skip, float* out)
{
    const int Ho = (H - 1) / stride + 1, Wo = (Wd - 1) / stride + 1, Po = Ho * Wo;
    int p = 0;
    for (; p + 4 <= Po; p += 4) { convChains<4>(cv, H, Wd, stride, Wo, in, skip, out, p, Po); }
    for (; p < Po; ++p) { convChains<1>(cv, H, Wd, stride, Wo, in, skip, out, p, Po); }
}
static void conv3x3s(const Conv& cv, int H, int Wd, int stride, const float* in, const float* skip, float* out)
{
    assert(cv.k == 3 && !cv.wk.empty());
    if (useAvx512()) { conv3x3s512(cv, H, Wd, stride, in, skip, out); } else { conv3x3sAvx2(cv, H, Wd, stride, in, skip, out); }
}
static void conv3x3(const Conv& cv, int H, int Wd, const float* in, const float* skip, float* out) { conv3x3s(cv, H, Wd, 1, in, skip, out); }
// The scalar statement of the same convolution — one fmaf per step, the accumulators in memory: what conv3x3s() was before its chains were register-blocked.  Kept as the
// definition convChains() is checked against on the CPU (tests/test_oracle_pinning.py::test_register_blocked_convolution_is_the_scalar_chain), not used by any forward.
static void conv3x3sScalar(const Conv& cv, int H, int Wd, int stride, const float* in, const float* skip, float* out)
{
    const int cin = cv.cin, cout = cv.cout, Ho = (H - 1) / stride + 1, Wo = (Wd - 1) / stride + 1, Pi = H * Wd, Po = Ho * Wo;
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x)
            for (int oc = 0; oc < cout; ++oc) {
                float acc = 0.0f;
                for (int t = 0; t < 9; ++t) {
                    const int yy = y * stride + t / 3 - 1, xx = x * stride + t % 3 - 1;
                    const bool inside = (yy >= 0 && yy < H && xx >= 0 && xx < Wd);
                    for (int c = 0; c < cin; ++c) {
                        const float xv = inside ? in[c * Pi + yy * Wd + xx] : 0.0f;
                        acc = __builtin_fmaf(xv, cv.w[(size_t(oc) * cin + c) * 9 + t], acc);
                    }
                }
                float v = acc + cv.b[oc];
                if (skip) { v = v + skip[oc * Po + y * Wo + x]; }
                out[oc * Po + y * Wo + x] = v > 0.0f ? v : 0.0f;
            }
}
// 0: every output bit of conv3x3s() equals the scalar chain's on a seeded random layer (weights and inputs in [-1, 1), a few exact zeros and denormal-sized values among them)
int convSelfTest(int cin, int cout, int H, int Wd, int stride, int with_skip, uint64_t seed)
{
    Conv c{cin, cout, 3, {}, {}};
    c.w.resize(size_t(cout) * cin * 9);
    c.b.resize(cout);
    uint64_t st = seed * 0x9E3779B97F4A7C15ULL + 12345;
    auto rnd = [&]() {
        st = mix64(st + 0x9E3779B97F4A7C15ULL);
        const unsigned r = unsigned(st >> 40);
        if ((r & 63) == 0) { return 0.0f; }
        const float u = float(r) * 5.9604644775390625e-08f * 2.0f - 1.0f;
        return (r & 63) == 1 ? u * 1e-38f : u;
    };
    for (auto& v : c.w) { v = rnd(); }
    for (auto& v : c.b) { v = rnd(); }
    finishConv(c);
    const int Ho = (H - 1) / stride + 1, Wo = (Wd - 1) / stride + 1;
    std::vector<float> in(size_t(cin) * H * Wd), sk(size_t(cout) * Ho * Wo), a(sk.size()), b(sk.size());
    for (auto& v : in) { v = rnd(); }
    for (auto& v : sk) { v = rnd(); }
    conv3x3sScalar(c, H, Wd, stride, in.data(), with_skip ? sk.data() : nullptr, b.data());
    conv3x3sAvx2(c, H, Wd, stride, in.data(), with_skip ? sk.data() : nullptr, a.data());
    int bad = memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0 ? 0 : 1;
    if (__builtin_cpu_supports("avx512f")) { // (both vector paths, whatever the dispatch would pick)
        std::fill(a.begin(), a.end(), -1.0f);
        conv3x3s512(c, H, Wd, stride, in.data(), with_skip ? sk.data() : nullptr, a.data());
        bad |= memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0 ? 0 : 2;
    }
    return bad;
}

static void conv1x1relu(const Conv& cv, int P, const float* in, float* out)
{
    for (int oc = 0; oc < cv.cout; ++oc)
        for (int p = 0; p < P; ++p) {
            float acc = 0.0f;
            for (int c = 0; c < cv.cin; ++c) { acc = __builtin_fmaf(in[c * P + p], cv.w[size_t(oc) * cv.cin + c], acc); }
            float v = acc + cv.b[oc];
            out[oc * P + p] = v > 0.0f ? v : 0.0f;
        }
}
static void linear(const Linear& l, const float* in, float* out, bool relu)
{
    for (int o = 0; o < l.out; ++o) {
        float acc = 0.0f;
        for (int i = 0; i < l.in; ++i) { acc = __builtin_fmaf(in[i], l.w[size_t(o) * l.in + i], acc); }
        float v = acc + l.b[o];
        out[o] = (relu && !(v > 0.0f)) ? 0.0f : v;
    }
}

struct Trunk {
    Conv stem;
    std::vector<Conv> blocks; // 2 per residual block
};
struct Heads {
    Conv pconv, vconv;
    Linear pfc, vfc1, vfc2;
};

class NetImpl : public Net {
public:
    Trunk repr, dyn;
    Heads heads;
    int H, Wd, P, C;

    static Trunk takeTrunk(const float*& p, int cin, const NetDesc& d)
    {
        Trunk t;
        t.stem = takeConvBN(p, cin, d.num_hidden_channels, 3);
        for (int b = 0; b < 2 * d.num_blocks; ++b) { t.blocks.push_back(takeConvBN(p, d.num_hidden_channels, d.num_hidden_channels, 3)); }
        return t;
    }
    void runTrunk(const Trunk& t, int h, int w, const float* in, float* x) const // ref alphazero_network.py:91-95, network_unit.py:14-23
    {
        std::vector<float> tmp(size_t(C) * h * w), y(size_t(C) * h * w);
        conv3x3(t.stem, h, w, in, nullptr, x);
        for (size_t b = 0; b < t.blocks.size(); b += 2) {
            conv3x3(t.blocks[b], h, w, x, nullptr, tmp.data());
            conv3x3(t.blocks[b + 1], h, w, tmp.data(), x, y.data());
            memcpy(x, y.data(), y.size() * sizeof(float));
        }
    }
    void runHeads(const float* x, float* policy, float* logit, float* value) const // ref network_unit.py:36-64, alphazero_network.py:97-104
    {
        const int A = desc.action_size, pc = heads.pconv.cout;
        std::vector<float> pf(size_t(pc) * P), vf(P), h1(desc.num_value_hidden_channels);
        conv1x1relu(heads.pconv, P, x, pf.data());
        linear(heads.pfc, pf.data(), logit, false);
        float m = logit[0];
        for (int a = 1; a < A; ++a) { m = logit[a] > m ? logit[a] : m; }
        float s = 0.0f;
        for (int a = 0; a < A; ++a) { policy[a] = mz_expf(logit[a] - m); s += policy[a]; }
        for (int a = 0; a < A; ++a) { policy[a] = policy[a] / s; }
        conv1x1relu(heads.vconv, P, x, vf.data());
        linear(heads.vfc1, vf.data(), h1.data(), true);
        float v;
        linear(heads.vfc2, h1.data(), &v, false);
        *value = mz_tanhf(v);
    }
    void scaleHidden(float* h) const // ref muzero_network.py:154-164
    {
        const int n = C * P;
        float mn = h[0], mx = h[0];
        for (int i = 1; i < n; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
        float scale = mx - mn;
        if (scale < 1e-5f) { scale += 1e-5f; }
        for (int i = 0; i < n; ++i) { h[i] = (h[i] - mn) / scale; }
    }
    template <class F>
    static void parallelFor(int n, F f)
    {
        std::function<void(int)> fn = f;
        NNPool::get().run(n, fn);
    }

    void forwardAZ(const float* features, int batch, float* policy, float* logit, float* value) const override
    {
        const int A = desc.action_size, fin = desc.num_input_channels * P;
        parallelFor(batch, [&](int b) {
            std::vector<float> x(size_t(C) * P);
            runTrunk(repr, H, Wd, features + size_t(b) * fin, x.data());
            runHeads(x.data(), policy + size_t(b) * A, logit + size_t(b) * A, value + b);
        });
    }
    void initialMZ(const float* features, int batch, float* policy, float* logit, float* value, float* hidden) const override
    { // ref muzero_network.py:137-143
        const int A = desc.action_size, fin = desc.num_input_channels * P;
        parallelFor(batch, [&](int b) {
            float* h = hidden + size_t(b) * C * P;
            runTrunk(repr, H, Wd, features + size_t(b) * fin, h);
            scaleHidden(h);
            runHeads(h, policy + size_t(b) * A, logit + size_t(b) * A, value + b);
        });
    }
    void recurrentMZ(const float* hidden_in, const float* action_plane, int batch, float* policy, float* logit, float* value, float* reward,
                     float* hidden_out) const override
    { // ref muzero_network.py:146-152, :31-38 (cat(hidden, action_plane) on the channel axis)
        const int A = desc.action_size, ac = desc.num_action_feature_channels;
        parallelFor(batch, [&](int b) {
            std::vector<float> in(size_t(C + ac) * P);
            memcpy(in.data(), hidden_in + size_t(b) * C * P, size_t(C) * P * sizeof(float));
            memcpy(in.data() + size_t(C) * P, action_plane + size_t(b) * ac * P, size_t(ac) * P * sizeof(float));
            float* h = hidden_out + size_t(b) * C * P;
            runTrunk(dyn, H, Wd, in.data(), h);
            scaleHidden(h);
            runHeads(h, policy + size_t(b) * A, logit + size_t(b) * A, value + b);
            if (reward) { reward[b] = 0.0f; }
        });
    }
};

// =====================================================================================
// muzero_atari — ref network/py/muzero_atari_network.py:7-198, network_unit.py:67-87,
// muzero_network.h:157-174 (601-bin decode), utils/utils.h:102-108 (invertValue)
// =====================================================================================
// AvgPool2d(kernel 3, stride 2, padding 1), count_include_pad: sum in (ky, kx) order, / 9
static void avgpool3s2(int C, int H, int Wd, const float* in, float* out)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (Wd - 1) / 2 + 1;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float acc = 0.0f;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        int yy = 2 * y + ky - 1, xx = 2 * x + kx - 1;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < Wd) { acc = acc + in[(c * H + yy) * Wd + xx]; }
                    }
                out[(c * Ho + y) * Wo + x] = acc / 9.0f;
            }
}
float invertValue(float value) // ref utils/utils.h:102-108 (inner part in double, powf in float)
{
    const float epsilon = 0.001;
    const float sign_value = (value > 0.0f ? 1.0f : (value == 0.0f ? 0.0f : -1.0f));
    return sign_value * (powf((::sqrt(1 + 4 * epsilon * (::fabs(static_cast<double>(value)) + 1 + epsilon)) - 1) / (2 * epsilon), 2.0f) - 1);
}
struct DiscreteHead { Conv conv; Linear fc1, fc2; };

class AtariNetImpl : public Net {
public:
    Conv conv1, conv2, dconv;
    std::vector<Conv> rb1, rb2, rb3, rblocks, dblocks; // two convs per residual block
    DiscreteHead reward, value;
    Conv pconv;
    Linear pfc;
    int C, H0, W0, h, w, P;

    static void takeRB(const float*& p, int ch, std::vector<Conv>& v) { v.push_back(takeConvBN(p, ch, ch, 3)); v.push_back(takeConvBN(p, ch, ch, 3)); }
    static DiscreteHead takeDiscrete(const float*& p, int C, int hw, int hidden, int size)
    {
        DiscreteHead d;
        int hc = (size + hw - 1) / hw;
        d.conv = takeConvBN(p, C, hc, 1);
        d.fc1 = takeLinear(p, hw * hc, hidden);
        d.fc2 = takeLinear(p, hidden, size);
        return d;
    }
    static void runRB(const std::vector<Conv>& v, size_t i, int H, int W, std::vector<float>& x)
    {
        std::vector<float> tmp(x.size()), y(x.size());
        conv3x3s(v[i], H, W, 1, x.data(), nullptr, tmp.data());
        conv3x3s(v[i + 1], H, W, 1, tmp.data(), x.data(), y.data());
        x.swap(y);
    }
    // expectation of the 601-bin softmax in the transformed space (ref muzero_network.h:157-162)
    float discreteExpectation(const DiscreteHead& dh, const float* x) const
    {
        const int hc = dh.conv.cout, size = dh.fc2.out;
        std::vector<float> f(size_t(hc) * P), h1(dh.fc1.out), lg(size);
        conv1x1relu(dh.conv, P, x, f.data());
        linear(dh.fc1, f.data(), h1.data(), true);
        linear(dh.fc2, h1.data(), lg.data(), false);
        float m = lg[0];
        for (int i = 1; i < size; ++i) { m = lg[i] > m ? lg[i] : m; }
        float s = 0.0f;
        for (int i = 0; i < size; ++i) { lg[i] = mz_expf(lg[i] - m); s += lg[i]; }
        float e = 0.0f;
        int start_value = -size / 2;
        for (int i = 0; i < size; ++i) { e = e + (lg[i] / s) * start_value++; }
        return e;
    }
    void policyHead(const float* x, float* policy, float* logit) const
    {
        const int A = desc.action_size;
        std::vector<float> pf(size_t(pconv.cout) * P);
        conv1x1relu(pconv, P, x, pf.data());
        linear(pfc, pf.data(), logit, false);
        float m = logit[0];
        for (int a = 1; a < A; ++a) { m = logit[a] > m ? logit[a] : m; }
        float s = 0.0f;
        for (int a = 0; a < A; ++a) { policy[a] = mz_expf(logit[a] - m); s += policy[a]; }
        for (int a = 0; a < A; ++a) { policy[a] = policy[a] / s; }
    }
    void scaleHidden(float* hd) const
    {
        const int n = C * P;
        float mn = hd[0], mx = hd[0];
        for (int i = 1; i < n; ++i) { mn = hd[i] < mn ? hd[i] : mn; mx = hd[i] > mx ? hd[i] : mx; }
        float scale = mx - mn;
        if (scale < 1e-5f) { scale += 1e-5f; }
        for (int i = 0; i < n; ++i) { hd[i] = (hd[i] - mn) / scale; }
    }
    template <class F>
    static void parallelFor(int n, F f)
    {
        std::function<void(int)> fn = f;
        NNPool::get().run(n, fn);
    }
    void forwardAZ(const float*, int, float*, float*, float*) const override {}
    void initialMZ(const float* features, int batch, float* policy, float* logit, float* value, float* hidden) const override
    { // ref muzero_atari_network.py:21-39,155-168
        const int A = desc.action_size, fin = desc.num_input_channels * H0 * W0;
        parallelFor(batch, [&](int b) {
            int H = H0, W = W0;
            std::vector<float> x(size_t(C / 2) * (H / 2) * (W / 2));
            conv3x3s(conv1, H, W, 2, features + size_t(b) * fin, nullptr, x.data());
            H = (H - 1) / 2 + 1; W = (W - 1) / 2 + 1;
            runRB(rb1, 0, H, W, x);
            std::vector<float> y(size_t(C) * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1));
            conv3x3s(conv2, H, W, 2, x.data(), nullptr, y.data());
            H = (H - 1) / 2 + 1; W = (W - 1) / 2 + 1;
            runRB(rb2, 0, H, W, y);
            std::vector<float> z(size_t(C) * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1));
            avgpool3s2(C, H, W, y.data(), z.data());
            H = (H - 1) / 2 + 1; W = (W - 1) / 2 + 1;
            runRB(rb3, 0, H, W, z);
            std::vector<float> u(size_t(C) * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1));
            avgpool3s2(C, H, W, z.data(), u.data());
            H = (H - 1) / 2 + 1; W = (W - 1) / 2 + 1;
            assert(H == h && W == w);
            for (size_t i = 0; i < rblocks.size(); i += 2) { runRB(rblocks, i, H, W, u); }
            float* hd = hidden + size_t(b) * C * P;
            memcpy(hd, u.data(), u.size() * sizeof(float));
            scaleHidden(hd);
            policyHead(hd, policy + size_t(b) * A, logit + size_t(b) * A);
            value[b] = invertValue(discreteExpectation(this->value, hd));
        });
    }
    void recurrentMZ(const float* hidden_in, const float* action_plane, int batch, float* policy, float* logit, float* value, float* reward_out,
                     float* hidden_out) const override
    { // ref muzero_atari_network.py:49-58,170-187: the reward head reads the UN-scaled next hidden state
        const int A = desc.action_size, ac = desc.num_action_feature_channels;
        parallelFor(batch, [&](int b) {
            std::vector<float> in(size_t(C + ac) * P), x(size_t(C) * P);
            memcpy(in.data(), hidden_in + size_t(b) * C * P, size_t(C) * P * sizeof(float));
            memcpy(in.data() + size_t(C) * P, action_plane + size_t(b) * ac * P, size_t(ac) * P * sizeof(float));
            conv3x3s(dconv, h, w, 1, in.data(), nullptr, x.data());
            for (size_t i = 0; i < dblocks.size(); i += 2) { runRB(dblocks, i, h, w, x); }
            if (reward_out) { reward_out[b] = invertValue(discreteExpectation(reward, x.data())); }
            float* hd = hidden_out + size_t(b) * C * P;
            memcpy(hd, x.data(), x.size() * sizeof(float));
            scaleHidden(hd);
            policyHead(hd, policy + size_t(b) * A, logit + size_t(b) * A);
            value[b] = invertValue(discreteExpectation(this->value, hd));
        });
    }
};

static std::unique_ptr<Net> createAtari(const NetDesc& d, const float* raw, size_t n)
{
    auto net = std::make_unique<AtariNetImpl>();
    net->desc = d;
    net->C = d.num_hidden_channels;
    net->H0 = d.input_channel_height; net->W0 = d.input_channel_width;
    net->h = d.hidden_channel_height; net->w = d.hidden_channel_width;
    net->P = net->h * net->w;
    const int C = net->C, hw = net->P;
    const float* p = raw;
    net->conv1 = takeConvBN(p, d.num_input_channels, C / 2, 3);
    AtariNetImpl::takeRB(p, C / 2, net->rb1);
    net->conv2 = takeConvBN(p, C / 2, C, 3);
    AtariNetImpl::takeRB(p, C, net->rb2);
    AtariNetImpl::takeRB(p, C, net->rb3);
    for (int b = 0; b < d.num_blocks; ++b) { AtariNetImpl::takeRB(p, C, net->rblocks); }
    net->dconv = takeConvBN(p, C + d.num_action_feature_channels, C, 3);
    for (int b = 0; b < d.num_blocks; ++b) { AtariNetImpl::takeRB(p, C, net->dblocks); }
    net->reward = AtariNetImpl::takeDiscrete(p, C, hw, C, d.discrete_value_size);
    int pc = policyChannels(d);
    net->pconv = takeConvBN(p, C, pc, 1);
    net->pfc = takeLinear(p, pc * hw, d.action_size);
    net->value = AtariNetImpl::takeDiscrete(p, C, hw, d.num_value_hidden_channels, d.discrete_value_size);
    if (p != raw + n) { return nullptr; }
    return net;
}

std::unique_ptr<Net> Net::create(const NetDesc& d, const float* raw, size_t n)
{
    if (n != rawParamCount(d)) { return nullptr; }
    if (d.type == 2) { return createAtari(d, raw, n); }
    auto net = std::make_unique<NetImpl>();
    net->desc = d;
    net->H = d.hidden_channel_height;
    net->Wd = d.hidden_channel_width;
    net->P = net->H * net->Wd;
    net->C = d.num_hidden_channels;
    const float* p = raw;
    net->repr = NetImpl::takeTrunk(p, d.num_input_channels, d);
    if (d.type == 1) { net->dyn = NetImpl::takeTrunk(p, d.num_hidden_channels + d.num_action_feature_channels, d); }
    const int hw = net->P, pc = policyChannels(d);
    net->heads.pconv = takeConvBN(p, d.num_hidden_channels, pc, 1);
    net->heads.pfc = takeLinear(p, pc * hw, d.action_size);
    net->heads.vconv = takeConvBN(p, d.num_hidden_channels, 1, 1);
    net->heads.vfc1 = takeLinear(p, hw, d.num_value_hidden_channels);
    net->heads.vfc2 = takeLinear(p, d.num_value_hidden_channels, 1);
    assert(p == raw + n);
    return net;
}

} // namespace mzo
