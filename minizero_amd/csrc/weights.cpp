// Host-side weight handling of libmzgpu: blob manifest (state_dict order of ref network/py/*.py modules),
// deterministic synthetic generator, eval-mode BatchNorm folding (ref network_unit.py:9-12, eps 1e-5) and
// packing into the layouts the gfx950 kernels read:
//   conv3x3  : wp[((t*CG + cg)*OT + ot)*64 + lane] = w'[16*ot + (lane&15)][4*cg + (lane>>4)][t]
//              (one coalesced 256-B read per wave = one 16x16x4 MFMA "A" fragment; zero padded)
//   linear   : transposed to [in][out] so that consecutive lanes (= outputs) read consecutive floats
#include "net.h"
#include <cmath>
#include <cstring>

namespace mz {

namespace {
enum Kind { W, Bv, BN_G, BN_B, BN_M, BN_V };
struct Spec { size_t n; int fan_in; Kind kind; };

int policyChannels(const mz_net_desc& d)
{
    int hw = d.hidden_channel_height * d.hidden_channel_width;
    return (d.action_size + hw - 1) / hw;
}
void convBN(std::vector<Spec>& m, int cin, int cout, int k)
{
    m.push_back({size_t(cout) * cin * k * k, cin * k * k, W});
    m.push_back({size_t(cout), cin * k * k, Bv});
    m.push_back({size_t(cout), 0, BN_G});
    m.push_back({size_t(cout), 0, BN_B});
    m.push_back({size_t(cout), 0, BN_M});
    m.push_back({size_t(cout), 0, BN_V});
}
void lin(std::vector<Spec>& m, int in, int out)
{
    m.push_back({size_t(out) * in, in, W});
    m.push_back({size_t(out), in, Bv});
}
std::vector<Spec> manifest(const mz_net_desc& d)
{
    std::vector<Spec> m;
    const int C = d.num_hidden_channels, hw = d.hidden_channel_height * d.hidden_channel_width;
    auto trunk = [&](int cin) {
        convBN(m, cin, C, 3);
        for (int b = 0; b < d.num_blocks; ++b) { convBN(m, C, C, 3); convBN(m, C, C, 3); }
    };
    int pc = policyChannels(d);
    if (d.type == 2) { // ref muzero_atari_network.py:7-70: representation, dynamics (+ reward head), prediction (policy, 601-bin value)
        auto rb = [&](int ch) { convBN(m, ch, ch, 3); convBN(m, ch, ch, 3); };
        auto discrete = [&](int hidden, int size) {
            int hc = (size + hw - 1) / hw;
            convBN(m, C, hc, 1);
            lin(m, hw * hc, hidden);
            lin(m, hidden, size);
        };
        convBN(m, d.num_input_channels, C / 2, 3);
        rb(C / 2);
        convBN(m, C / 2, C, 3);
        rb(C);
        rb(C);
        for (int b = 0; b < d.num_blocks; ++b) { rb(C); }
        convBN(m, C + d.num_action_feature_channels, C, 3);
        for (int b = 0; b < d.num_blocks; ++b) { rb(C); }
        discrete(C, d.discrete_value_size);
        convBN(m, C, pc, 1);
        lin(m, pc * hw, d.action_size);
        discrete(d.num_value_hidden_channels, d.discrete_value_size);
        return m;
    }
    trunk(d.num_input_channels);
    if (d.type == 1) { trunk(C + d.num_action_feature_channels); }
    convBN(m, C, pc, 1);
    lin(m, pc * hw, d.action_size);
    convBN(m, C, 1, 1);
    lin(m, hw, d.num_value_hidden_channels);
    lin(m, d.num_value_hidden_channels, 1);
    return m;
}
inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
} // namespace

bool netValidateDesc(const mz_net_desc& d)
{
    if (d.type < 0 || d.type > 2) { setError("unknown network type %d (alphazero=0, muzero=1, muzero_atari=2)", d.type); return false; }
    if (d.type == 2) {
        if (d.discrete_value_size < 3 || d.input_channel_height != 16 * d.hidden_channel_height || d.input_channel_width != 16 * d.hidden_channel_width ||
            d.num_hidden_channels % 32 != 0) {
            setError("muzero_atari needs discrete_value_size >= 3, input = 16 x hidden resolution and hidden channels %% 32 == 0");
            return false;
        }
    } else {
        if (d.discrete_value_size != 1) { setError("discrete_value_size %d is only supported by the muzero_atari network", d.discrete_value_size); return false; }
        if (d.input_channel_height != d.hidden_channel_height || d.input_channel_width != d.hidden_channel_width) {
            setError("input and hidden planes must have the same size (board games)");
            return false;
        }
    }
    if (d.num_hidden_channels <= 0 || d.num_blocks < 0 || d.action_size <= 0 || d.num_input_channels <= 0) { setError("bad network descriptor"); return false; }
    return true;
}

long netParamCount(const mz_net_desc& d)
{
    size_t n = 0;
    for (auto& s : manifest(d)) { n += s.n; }
    return static_cast<long>(n);
}

bool netGenerate(const mz_net_desc& d, uint64_t seed, float* out)
{
    size_t idx = 0;
    for (auto& s : manifest(d)) {
        float lo, hi;
        if (s.kind == W || s.kind == Bv) {
            float bound = 1.0f / sqrtf(static_cast<float>(s.fan_in));
            lo = -bound;
            hi = bound;
        } else if (s.kind == BN_G || s.kind == BN_V) {
            lo = 0.5f;
            hi = 1.5f;
        } else {
            lo = -0.1f;
            hi = 0.1f;
        }
        for (size_t i = 0; i < s.n; ++i, ++idx) {
            uint64_t z = mix64(seed + (idx + 1) * 0x9E3779B97F4A7C15ULL);
            float u = static_cast<float>(z >> 40) * 5.9604644775390625e-08f;
            out[idx] = lo + (hi - lo) * u;
        }
    }
    return true;
}

// ---- folding + packing ----
namespace {
struct Folded { int cin, cout, k; std::vector<float> w, b; };

Folded takeConvBN(const float*& p, int cin, int cout, int k)
{
    Folded f{cin, cout, k, {}, {}};
    const size_t per = size_t(cin) * k * k, nw = per * cout;
    const float *w = p, *b = p + nw, *g = b + cout, *be = g + cout, *mu = be + cout, *var = mu + cout;
    p = var + cout;
    f.w.resize(nw);
    f.b.resize(cout);
    for (int oc = 0; oc < cout; ++oc) {
        const float s = g[oc] / sqrtf(var[oc] + 1e-5f);
        for (size_t i = 0; i < per; ++i) { f.w[oc * per + i] = w[oc * per + i] * s; }
        const float t = (b[oc] - mu[oc]) * s;
        f.b[oc] = t + be[oc];
    }
    return f;
}

size_t append(std::vector<float>& dst, const std::vector<float>& src)
{
    // keep every tensor 64-float (256 B) aligned so wave-wide reads stay on cache-line boundaries
    while (dst.size() % 64) { dst.push_back(0.0f); }
    size_t off = dst.size();
    dst.insert(dst.end(), src.begin(), src.end());
    return off;
}

ConvLayer packConv3(std::vector<float>& dst, const Folded& f)
{
    ConvLayer L;
    L.cin = f.cin;
    L.cin_pad = (f.cin + 3) / 4 * 4;
    L.cout = f.cout;
    L.cout_pad = (f.cout + 15) / 16 * 16;
    const int CG = L.cin_pad / 4, OT = L.cout_pad / 16;
    std::vector<float> wp(size_t(9) * CG * OT * 64, 0.0f);
    for (int t = 0; t < 9; ++t)
        for (int cg = 0; cg < CG; ++cg)
            for (int ot = 0; ot < OT; ++ot)
                for (int l = 0; l < 64; ++l) {
                    int oc = 16 * ot + (l & 15), c = 4 * cg + (l >> 4);
                    if (oc < f.cout && c < f.cin) { wp[((size_t(t) * CG + cg) * OT + ot) * 64 + l] = f.w[(size_t(oc) * f.cin + c) * 9 + t]; }
                }
    L.w_off = append(dst, wp);
    // fused-tower layout: for (tap, oc-tile) the CG fragments are contiguous; groups of four channel groups are interleaved per lane so
    // that one global_load_dwordx4 brings a lane its A values of four k-steps; the CG % 4 remaining fragments follow as plain 64-float rows
    {
        std::vector<float> w4(wp.size(), 0.0f);
        const int CG4 = CG / 4;
        for (int t = 0; t < 9; ++t)
            for (int ot = 0; ot < OT; ++ot)
                for (int cg = 0; cg < CG; ++cg)
                    for (int l = 0; l < 64; ++l) {
                        const size_t base = (size_t(t) * OT + ot) * CG * 64;
                        const size_t idx = cg < 4 * CG4 ? base + size_t(cg >> 2) * 256 + size_t(l) * 4 + (cg & 3)
                                                        : base + size_t(CG4) * 256 + size_t(cg - 4 * CG4) * 64 + l;
                        w4[idx] = wp[((size_t(t) * CG + cg) * OT + ot) * 64 + l];
                    }
        L.w4_off = append(dst, w4);
    }
    // wide-tower layout (net_wide_body.h): whole dwordx4 chunks only — the input channels padded to a multiple of 16 (zero weights: an fma with a zero
    // operand leaves the chain's value as it is); for layers whose channels are a multiple of 16 that is the w4 layout itself
    L.cq = (f.cin + 15) / 16;
    if (L.cin_pad % 16 == 0) {
        L.wq_off = L.w4_off;
    } else {
        std::vector<float> wq(size_t(9) * OT * L.cq * 256, 0.0f);
        for (int t = 0; t < 9; ++t)
            for (int ot = 0; ot < OT; ++ot)
                for (int cg = 0; cg < CG; ++cg)
                    for (int l = 0; l < 64; ++l) {
                        wq[((size_t(t) * OT + ot) * L.cq + (cg >> 2)) * 256 + size_t(l) * 4 + (cg & 3)] = wp[((size_t(t) * CG + cg) * OT + ot) * 64 + l];
                    }
        L.wq_off = append(dst, wq);
    }
    std::vector<float> b(L.cout_pad, 0.0f);
    for (int oc = 0; oc < f.cout; ++oc) { b[oc] = f.b[oc]; }
    L.b_off = append(dst, b);
    return L;
}

std::vector<ConvLayer> packTrunk(std::vector<float>& dst, const float*& p, int cin, const mz_net_desc& d)
{
    std::vector<ConvLayer> t;
    t.push_back(packConv3(dst, takeConvBN(p, cin, d.num_hidden_channels, 3)));
    for (int b = 0; b < 2 * d.num_blocks; ++b) { t.push_back(packConv3(dst, takeConvBN(p, d.num_hidden_channels, d.num_hidden_channels, 3))); }
    return t;
}

std::vector<float> transposeLinear(const float* w, int in, int out)
{
    std::vector<float> t(size_t(in) * out);
    for (int o = 0; o < out; ++o)
        for (int i = 0; i < in; ++i) { t[size_t(i) * out + o] = w[size_t(o) * in + i]; }
    return t;
}
} // namespace

namespace {
DiscreteHeadOffsets packDiscrete(std::vector<float>& packed, const float*& p, int C, int hw, int hidden, int size)
{
    DiscreteHeadOffsets o;
    o.hc = (size + hw - 1) / hw;
    o.hidden = hidden;
    o.size = size;
    Folded conv = takeConvBN(p, C, o.hc, 1);
    o.conv_w = append(packed, conv.w); // [hc][C]
    o.conv_b = append(packed, conv.b);
    o.fc1_wT = append(packed, transposeLinear(p, hw * o.hc, hidden));
    p += size_t(hw) * o.hc * hidden;
    o.fc1_b = append(packed, std::vector<float>(p, p + hidden));
    p += hidden;
    o.fc2_wT = append(packed, transposeLinear(p, hidden, size));
    p += size_t(hidden) * size;
    o.fc2_b = append(packed, std::vector<float>(p, p + size));
    p += size;
    return o;
}
void packRB(std::vector<float>& packed, const float*& p, int ch, std::vector<ConvLayer>& v)
{
    v.push_back(packConv3(packed, takeConvBN(p, ch, ch, 3)));
    v.push_back(packConv3(packed, takeConvBN(p, ch, ch, 3)));
}
} // namespace

bool packWeights(const mz_net_desc& d, const float* raw, size_t n, std::vector<float>& packed, std::vector<ConvLayer>& repr, std::vector<ConvLayer>& dyn,
                 HeadOffsets& h, AtariLayers& at)
{
    if (d.type == 2) {
        if (static_cast<long>(n) != netParamCount(d)) {
            setError("weight blob has %zu floats, descriptor needs %ld", n, netParamCount(d));
            return false;
        }
        packed.clear();
        repr.clear();
        dyn.clear();
        at = AtariLayers();
        const float* p = raw;
        const int C = d.num_hidden_channels, hw = d.hidden_channel_height * d.hidden_channel_width;
        at.conv1 = packConv3(packed, takeConvBN(p, d.num_input_channels, C / 2, 3));
        packRB(packed, p, C / 2, at.rb1);
        at.conv2 = packConv3(packed, takeConvBN(p, C / 2, C, 3));
        packRB(packed, p, C, at.rb2);
        packRB(packed, p, C, at.rb3);
        for (int b = 0; b < d.num_blocks; ++b) { packRB(packed, p, C, at.tail); }
        dyn.push_back(packConv3(packed, takeConvBN(p, C + d.num_action_feature_channels, C, 3)));
        for (int b = 0; b < d.num_blocks; ++b) { packRB(packed, p, C, dyn); }
        at.reward = packDiscrete(packed, p, C, hw, C, d.discrete_value_size);
        h.pc = policyChannels(d);
        Folded pconv = takeConvBN(p, C, h.pc, 1);
        h.pconv_w = append(packed, pconv.w);
        h.pconv_b = append(packed, pconv.b);
        h.pfc_wT = append(packed, transposeLinear(p, h.pc * hw, d.action_size));
        p += size_t(h.pc) * hw * d.action_size;
        h.pfc_b = append(packed, std::vector<float>(p, p + d.action_size));
        p += d.action_size;
        at.value = packDiscrete(packed, p, C, hw, d.num_value_hidden_channels, d.discrete_value_size);
        if (p != raw + n) { setError("internal: atari weight manifest mismatch"); return false; }
        return true;
    }
    if (static_cast<long>(n) != netParamCount(d)) {
        setError("weight blob has %zu floats, descriptor needs %ld", n, netParamCount(d));
        return false;
    }
    packed.clear();
    const float* p = raw;
    const int C = d.num_hidden_channels, hw = d.hidden_channel_height * d.hidden_channel_width;
    repr = packTrunk(packed, p, d.num_input_channels, d);
    dyn.clear();
    if (d.type == 1) { dyn = packTrunk(packed, p, C + d.num_action_feature_channels, d); }
    h.pc = policyChannels(d);
    Folded pconv = takeConvBN(p, C, h.pc, 1);
    h.pconv_w = append(packed, pconv.w); // [pc][C]
    h.pconv_b = append(packed, pconv.b);
    h.pfc_wT = append(packed, transposeLinear(p, h.pc * hw, d.action_size));
    p += size_t(h.pc) * hw * d.action_size;
    h.pfc_b = append(packed, std::vector<float>(p, p + d.action_size));
    p += d.action_size;
    Folded vconv = takeConvBN(p, C, 1, 1);
    h.vconv_w = append(packed, vconv.w); // [C]
    h.vconv_b = append(packed, vconv.b);
    const int VH = d.num_value_hidden_channels;
    h.vfc1_wT = append(packed, transposeLinear(p, hw, VH));
    p += size_t(hw) * VH;
    h.vfc1_b = append(packed, std::vector<float>(p, p + VH));
    p += VH;
    h.vfc2_w = append(packed, std::vector<float>(p, p + VH));
    p += VH;
    h.vfc2_b = append(packed, std::vector<float>(p, p + 1));
    p += 1;
    if (p != raw + n) { setError("internal: weight manifest mismatch"); return false; }
    return true;
}

} // namespace mz
