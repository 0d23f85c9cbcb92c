// Native reader of the TorchScript model files the reference's trainer writes (ref learner/train.py:127
// `torch.jit.script(self.network.module).save(".../weight_iter_N.pt")`, loaded by Network::loadModel with torch::jit::load,
// ref network/network.h:18-37).  No LibTorch: a .pt file is a ZIP archive of STORED (uncompressed) members —
//   <name>/data.pkl      pickle (protocol 2) of the scripted module: nested objects whose attribute dicts hold the hyper-parameters
//                        (game_name, num_blocks, ... — the values behind get_num_blocks() etc., ref alphazero_network.py:42-88) and
//                        the tensors as torch._utils._rebuild_tensor_v2(persistent-id storage, offset, size, stride, ...)
//   <name>/data/<key>    raw little-endian storages
// What comes out is what mz_net_create takes: the 12 hyper-parameters and every floating tensor in attribute order, which for
// the reference's modules is state_dict() order (num_batches_tracked is an int64 tensor and drops out) — the layout of
// minizero_amd/export_weights.py; tests/test_pt_reader.py checks the two against each other on files scripted from the
// reference's own Python modules.
#include "common.h"
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace mz {

namespace {

struct ZipEntry { uint64_t data_off = 0, size = 0; int method = 0; };

uint32_t rd32(const uint8_t* p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }
uint16_t rd16(const uint8_t* p) { return uint16_t(p[0] | p[1] << 8); }
uint64_t rd64(const uint8_t* p) { return uint64_t(rd32(p)) | uint64_t(rd32(p + 4)) << 32; }

// central directory of a ZIP / ZIP64 archive -> member name -> (offset of the member's bytes, size, method)
bool readZipDirectory(const std::vector<uint8_t>& f, std::map<std::string, ZipEntry>& out, std::string& err)
{
    if (f.size() < 22) { err = "not a zip archive (too small)"; return false; }
    size_t eocd = std::string::npos;
    for (size_t i = f.size() - 22;; --i) {
        if (rd32(&f[i]) == 0x06054b50u) { eocd = i; break; }
        if (i == 0 || f.size() - i > 70000) { break; }
    }
    if (eocd == std::string::npos) { err = "not a zip archive (no end-of-central-directory record)"; return false; }
    uint64_t count = rd16(&f[eocd + 10]), cd_size = rd32(&f[eocd + 12]), cd_off = rd32(&f[eocd + 16]);
    if (count == 0xFFFF || cd_off == 0xFFFFFFFFu || cd_size == 0xFFFFFFFFu) { // ZIP64: locator right before the EOCD
        if (eocd < 20 || rd32(&f[eocd - 20]) != 0x07064b50u) { err = "zip64 locator missing"; return false; }
        const uint64_t e64 = rd64(&f[eocd - 20 + 8]);
        if (e64 > f.size() || f.size() - e64 < 56 || rd32(&f[e64]) != 0x06064b50u) { err = "zip64 end record missing"; return false; }
        count = rd64(&f[e64 + 32]); cd_size = rd64(&f[e64 + 40]); cd_off = rd64(&f[e64 + 48]);
    }
    if (cd_off > f.size() || cd_size > f.size() - cd_off) { err = "zip central directory out of range"; return false; }
    size_t p = cd_off;
    for (uint64_t i = 0; i < count; ++i) {
        if (p + 46 > f.size() || rd32(&f[p]) != 0x02014b50u) { err = "bad central directory entry"; return false; }
        const int method = rd16(&f[p + 10]);
        uint64_t csize = rd32(&f[p + 20]), usize = rd32(&f[p + 24]), lho = rd32(&f[p + 42]);
        const size_t nlen = rd16(&f[p + 28]), xlen = rd16(&f[p + 30]), clen = rd16(&f[p + 32]);
        if (p + 46 + nlen + xlen > f.size()) { err = "bad central directory entry"; return false; }
        const std::string name(reinterpret_cast<const char*>(&f[p + 46]), nlen);
        size_t x = p + 46 + nlen;
        const size_t xend = x + xlen;
        while (x + 4 <= xend) { // ZIP64 extended information
            const int id = rd16(&f[x]), sz = rd16(&f[x + 2]);
            if (id == 1) {
                size_t q = x + 4;
                if (usize == 0xFFFFFFFFu && q + 8 <= xend) { usize = rd64(&f[q]); q += 8; }
                if (csize == 0xFFFFFFFFu && q + 8 <= xend) { csize = rd64(&f[q]); q += 8; }
                if (lho == 0xFFFFFFFFu && q + 8 <= xend) { lho = rd64(&f[q]); q += 8; }
            }
            x += 4 + sz;
        }
        // offsets and sizes come from the file: compare without additions that could wrap (crafted ZIP64 values)
        if (lho > f.size() || f.size() - lho < 30 || rd32(&f[lho]) != 0x04034b50u) { err = "bad local header for " + name; return false; }
        ZipEntry e;
        e.data_off = lho + 30 + rd16(&f[lho + 26]) + rd16(&f[lho + 28]);
        e.size = usize;
        e.method = method;
        const uint64_t stored = method == 0 ? usize : csize;
        if (e.data_off > f.size() || stored > f.size() - e.data_off) { err = "member out of range: " + name; return false; }
        out[name] = e;
        p += 46 + nlen + xlen + clen;
    }
    return true;
}

// ---- the subset of pickle protocol 2 that torch.jit's pickler emits ----
struct PV;
using PVP = std::shared_ptr<PV>;
struct PV {
    enum Kind { None, Bool, Int, Float, Str, Tuple, List, Dict, Global, Object, Storage, Tensor } kind = None;
    int64_t i = 0;
    double f = 0;
    std::string s;                           // Str / Global ("module name") / Storage key
    std::vector<PVP> items;                  // Tuple / List / Object constructor args
    std::vector<std::pair<PVP, PVP>> dict;   // Dict, insertion order
    PVP cls, state;                          // Object
    bool is_f32 = false;                     // Storage / Tensor
    int64_t offset = 0, storage_numel = 0;
    std::vector<int64_t> sizes, strides;
};
PVP mk(PV::Kind k) { auto p = std::make_shared<PV>(); p->kind = k; return p; }

class Unpickler {
public:
    Unpickler(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    bool run(PVP& root, std::string& err)
    {
        while (true) {
            if (pos_ >= n_) { err = "pickle: unexpected end"; return false; }
            const uint8_t op = p_[pos_++];
            switch (op) {
            case 0x80: if (!need(1, err)) { return false; } ++pos_; break;                                  // PROTO
            case '.': if (stack_.empty()) { err = "pickle: empty stack at STOP"; return false; } root = stack_.back(); return true;
            case 'c': { // GLOBAL
                std::string mod, name;
                if (!line(mod, err) || !line(name, err)) { return false; }
                auto g = mk(PV::Global); g->s = mod + " " + name; stack_.push_back(g);
                break;
            }
            case 'q': if (!need(1, err) || !top(err)) { return false; } memo_[p_[pos_++]] = stack_.back(); break;      // BINPUT
            case 'r': if (!need(4, err) || !top(err)) { return false; } memo_[rd32(p_ + pos_)] = stack_.back(); pos_ += 4; break; // LONG_BINPUT
            case 'h': { if (!need(1, err)) { return false; } if (!get(p_[pos_++], err)) { return false; } break; }       // BINGET
            case 'j': { if (!need(4, err)) { return false; } const uint32_t k = rd32(p_ + pos_); pos_ += 4; if (!get(k, err)) { return false; } break; }
            case ')': stack_.push_back(mk(PV::Tuple)); break;
            case ']': stack_.push_back(mk(PV::List)); break;
            case '}': stack_.push_back(mk(PV::Dict)); break;
            case '(': marks_.push_back(stack_.size()); break;
            case 'N': stack_.push_back(mk(PV::None)); break;
            case 0x88: { auto b = mk(PV::Bool); b->i = 1; stack_.push_back(b); break; }
            case 0x89: { auto b = mk(PV::Bool); b->i = 0; stack_.push_back(b); break; }
            case 'K': { if (!need(1, err)) { return false; } pushInt(p_[pos_]); pos_ += 1; break; }
            case 'M': { if (!need(2, err)) { return false; } pushInt(rd16(p_ + pos_)); pos_ += 2; break; }
            case 'J': { if (!need(4, err)) { return false; } pushInt(static_cast<int32_t>(rd32(p_ + pos_))); pos_ += 4; break; }
            case 0x8a: { // LONG1
                if (!need(1, err)) { return false; }
                const size_t len = p_[pos_++];
                if (!need(len, err) || len > 8) { err = "pickle: LONG1 too long"; return false; }
                uint64_t v = 0;
                for (size_t k = 0; k < len; ++k) { v |= uint64_t(p_[pos_ + k]) << (8 * k); }
                if (len > 0 && len < 8 && (p_[pos_ + len - 1] & 0x80)) { v |= ~uint64_t(0) << (8 * len); }
                pos_ += len;
                pushInt(static_cast<int64_t>(v));
                break;
            }
            case 'G': { // BINFLOAT, big-endian
                if (!need(8, err)) { return false; }
                uint64_t v = 0;
                for (int k = 0; k < 8; ++k) { v = v << 8 | p_[pos_ + k]; }
                pos_ += 8;
                auto f = mk(PV::Float); memcpy(&f->f, &v, 8); stack_.push_back(f);
                break;
            }
            case 'X': { // BINUNICODE
                if (!need(4, err)) { return false; }
                const uint32_t len = rd32(p_ + pos_); pos_ += 4;
                if (!need(len, err)) { return false; }
                auto s = mk(PV::Str); s->s.assign(reinterpret_cast<const char*>(p_ + pos_), len); pos_ += len; stack_.push_back(s);
                break;
            }
            case 't': { if (!popMark(PV::Tuple, err)) { return false; } break; }
            case 0x85: case 0x86: case 0x87: {
                const size_t k = op - 0x84;
                if (stack_.size() < k) { err = "pickle: stack underflow"; return false; }
                auto t = mk(PV::Tuple); t->items.assign(stack_.end() - k, stack_.end()); stack_.resize(stack_.size() - k); stack_.push_back(t);
                break;
            }
            case 'l': { if (!popMark(PV::List, err)) { return false; } break; }
            case 'a': { // APPEND
                if (stack_.size() < 2) { err = "pickle: stack underflow"; return false; }
                PVP v = stack_.back(); stack_.pop_back();
                stack_.back()->items.push_back(v);
                break;
            }
            case 'e': { // APPENDS
                if (marks_.empty()) { err = "pickle: no mark"; return false; }
                const size_t m = marks_.back(); marks_.pop_back();
                if (m == 0 || m > stack_.size()) { err = "pickle: bad mark"; return false; }
                PVP l = stack_[m - 1];
                l->items.insert(l->items.end(), stack_.begin() + m, stack_.end());
                stack_.resize(m);
                break;
            }
            case 's': { // SETITEM
                if (stack_.size() < 3) { err = "pickle: stack underflow"; return false; }
                PVP v = stack_.back(); stack_.pop_back();
                PVP k = stack_.back(); stack_.pop_back();
                stack_.back()->dict.emplace_back(k, v);
                break;
            }
            case 'u': { // SETITEMS
                if (marks_.empty()) { err = "pickle: no mark"; return false; }
                const size_t m = marks_.back(); marks_.pop_back();
                if (m == 0 || m > stack_.size() || (stack_.size() - m) % 2) { err = "pickle: bad SETITEMS"; return false; }
                PVP d = stack_[m - 1];
                for (size_t k = m; k + 1 < stack_.size(); k += 2) { d->dict.emplace_back(stack_[k], stack_[k + 1]); }
                stack_.resize(m);
                break;
            }
            case 0x81: { // NEWOBJ: cls, args
                if (stack_.size() < 2) { err = "pickle: stack underflow"; return false; }
                PVP args = stack_.back(); stack_.pop_back();
                PVP cls = stack_.back(); stack_.pop_back();
                auto o = mk(PV::Object); o->cls = cls; o->items = args->items; stack_.push_back(o);
                break;
            }
            case 'b': { // BUILD: obj, state
                if (stack_.size() < 2) { err = "pickle: stack underflow"; return false; }
                PVP st = stack_.back(); stack_.pop_back();
                if (stack_.back()->kind == PV::Object) { stack_.back()->state = st; }
                break;
            }
            case 'Q': { // BINPERSID: ('storage', <storage type>, key, location, numel)
                if (!top(err)) { return false; }
                PVP pid = stack_.back(); stack_.pop_back();
                if (pid->kind != PV::Tuple || pid->items.size() < 5 || pid->items[0]->kind != PV::Str || pid->items[0]->s != "storage") {
                    err = "pickle: unsupported persistent id"; return false;
                }
                auto st = mk(PV::Storage);
                st->is_f32 = pid->items[1]->kind == PV::Global && pid->items[1]->s == "torch FloatStorage";
                st->s = pid->items[2]->s;
                st->storage_numel = pid->items[4]->i;
                stack_.push_back(st);
                break;
            }
            case 'R': { // REDUCE: callable, args
                if (stack_.size() < 2) { err = "pickle: stack underflow"; return false; }
                PVP args = stack_.back(); stack_.pop_back();
                PVP fn = stack_.back(); stack_.pop_back();
                const std::string name = fn->kind == PV::Global ? fn->s : std::string();
                if (name == "torch._utils _rebuild_tensor_v2" && args->items.size() >= 4 && args->items[0]->kind == PV::Storage) {
                    auto t = mk(PV::Tensor);
                    t->s = args->items[0]->s; t->is_f32 = args->items[0]->is_f32; t->storage_numel = args->items[0]->storage_numel;
                    t->offset = args->items[1]->i;
                    for (auto& d : args->items[2]->items) { t->sizes.push_back(d->i); }
                    for (auto& d : args->items[3]->items) { t->strides.push_back(d->i); }
                    stack_.push_back(t);
                } else if (name == "torch._utils _rebuild_parameter" && !args->items.empty()) {
                    stack_.push_back(args->items[0]);
                } else if (name == "collections OrderedDict") {
                    stack_.push_back(mk(PV::Dict));
                } else {
                    auto o = mk(PV::Object); o->cls = fn; o->items = args->items; stack_.push_back(o);
                }
                break;
            }
            default: {
                char b[64];
                snprintf(b, sizeof(b), "pickle: unsupported opcode 0x%02x at %zu", op, pos_ - 1);
                err = b;
                return false;
            }
            }
        }
    }

private:
    bool need(size_t k, std::string& err) { if (pos_ + k > n_) { err = "pickle: truncated"; return false; } return true; }
    bool top(std::string& err) { if (stack_.empty()) { err = "pickle: stack underflow"; return false; } return true; }
    bool line(std::string& s, std::string& err)
    {
        const size_t b = pos_;
        while (pos_ < n_ && p_[pos_] != '\n') { ++pos_; }
        if (pos_ >= n_) { err = "pickle: unterminated GLOBAL"; return false; }
        s.assign(reinterpret_cast<const char*>(p_ + b), pos_ - b);
        ++pos_;
        return true;
    }
    bool get(uint32_t k, std::string& err)
    {
        auto it = memo_.find(k);
        if (it == memo_.end()) { err = "pickle: memo miss"; return false; }
        stack_.push_back(it->second);
        return true;
    }
    void pushInt(int64_t v) { auto i = mk(PV::Int); i->i = v; stack_.push_back(i); }
    bool popMark(PV::Kind k, std::string& err)
    {
        if (marks_.empty()) { err = "pickle: no mark"; return false; }
        const size_t m = marks_.back(); marks_.pop_back();
        if (m > stack_.size()) { err = "pickle: bad mark"; return false; }
        auto t = mk(k); t->items.assign(stack_.begin() + m, stack_.end()); stack_.resize(m); stack_.push_back(t);
        return true;
    }
    const uint8_t* p_;
    size_t n_, pos_ = 0;
    std::vector<PVP> stack_;
    std::vector<size_t> marks_;
    std::map<uint32_t, PVP> memo_;
};

const PV* attr(const PV& obj, const char* name)
{
    if (obj.kind != PV::Object || !obj.state || obj.state->kind != PV::Dict) { return nullptr; }
    for (auto& kv : obj.state->dict) { if (kv.first->kind == PV::Str && kv.first->s == name) { return kv.second.get(); } }
    return nullptr;
}

// depth-first over the attribute dicts in insertion order: every f32 tensor, made contiguous
bool collect(const PV& v, const std::vector<uint8_t>& file, const std::map<std::string, ZipEntry>& zip, const std::string& prefix,
             std::vector<float>& out, std::string& err)
{
    if (v.kind == PV::Tensor) {
        if (!v.is_f32) { return true; } // num_batches_tracked (int64)
        auto it = zip.find(prefix + "data/" + v.s);
        if (it == zip.end() || it->second.method != 0) { err = "storage " + v.s + " missing or compressed"; return false; }
        const float* base = reinterpret_cast<const float*>(&file[it->second.data_off]);
        const int64_t avail = static_cast<int64_t>(it->second.size / 4);
        if (v.strides.size() != v.sizes.size()) { err = "tensor with " + std::to_string(v.sizes.size()) + " sizes but " + std::to_string(v.strides.size()) + " strides"; return false; }
        int64_t numel = 1;
        for (int64_t d : v.sizes) {
            if (d < 0 || (d > 0 && numel > avail / d)) { err = "tensor larger than its storage"; return false; } // also bounds stride-0 views
            numel *= d;
        }
        if (numel > avail || out.size() + static_cast<size_t>(numel) > (size_t(1) << 31)) { err = "tensor larger than its storage"; return false; }
        std::vector<int64_t> idx(v.sizes.size(), 0);
        for (int64_t k = 0; k < numel; ++k) {
            int64_t off = v.offset;
            for (size_t d = 0; d < idx.size(); ++d) { off += idx[d] * v.strides[d]; }
            if (off < 0 || off >= avail) { err = "tensor view out of its storage"; return false; }
            float x;
            memcpy(&x, base + off, 4);
            out.push_back(x);
            for (int d = static_cast<int>(idx.size()) - 1; d >= 0; --d) { if (++idx[d] < v.sizes[d]) { break; } idx[d] = 0; }
        }
        return true;
    }
    if (v.kind == PV::Object && v.state && v.state->kind == PV::Dict) {
        for (auto& kv : v.state->dict) { if (!collect(*kv.second, file, zip, prefix, out, err)) { return false; } }
    }
    return true;
}

} // namespace

// desc + flat weights (the mz_net_create layout) of a TorchScript file written by the reference's trainer
bool readTorchScript(const std::string& path, mz_net_desc* desc, std::vector<float>* weights, std::string* err_out)
{
    std::string err;
    auto fail = [&](const std::string& m) { if (err_out) { *err_out = path + ": " + m; } return false; };
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) { return fail("cannot open"); }
    const std::streamsize n = in.tellg();
    if (n <= 0) { return fail("empty file"); }
    std::vector<uint8_t> file(static_cast<size_t>(n));
    in.seekg(0);
    if (!in.read(reinterpret_cast<char*>(file.data()), n)) { return fail("read error"); }
    std::map<std::string, ZipEntry> zip;
    if (!readZipDirectory(file, zip, err)) { return fail(err); }
    std::string prefix;
    const ZipEntry* pkl = nullptr;
    for (auto& kv : zip) {
        const std::string& name = kv.first;
        if (name.size() >= 8 && name.compare(name.size() - 8, 8, "data.pkl") == 0 && (name.size() == 8 || name[name.size() - 9] == '/')) {
            pkl = &kv.second;
            prefix = name.substr(0, name.size() - 8);
        }
    }
    if (!pkl || pkl->method != 0) { return fail("no stored data.pkl member (not a TorchScript archive?)"); }
    PVP root;
    Unpickler up(&file[pkl->data_off], pkl->size);
    if (!up.run(root, err)) { return fail(err); }
    if (!root || root->kind != PV::Object || !root->cls || root->cls->kind != PV::Global) { return fail("data.pkl does not hold a module object"); }
    const std::string& cls = root->cls->s;
    mz_net_desc d;
    memset(&d, 0, sizeof(d));
    d.type = cls.find("MuZeroAtariNetwork") != std::string::npos ? 2 : (cls.find("MuZeroNetwork") != std::string::npos ? 1
             : (cls.find("AlphaZeroNetwork") != std::string::npos ? 0 : -1));
    if (d.type < 0) { return fail("unknown network class " + cls); }
    const PV* g = attr(*root, "game_name");
    if (!g || g->kind != PV::Str) { return fail("attribute game_name missing"); }
    snprintf(d.game_name, sizeof(d.game_name), "%s", g->s.c_str());
    struct { const char* name; int* dst; bool required; } ints[] = {
        {"num_input_channels", &d.num_input_channels, true}, {"input_channel_height", &d.input_channel_height, true},
        {"input_channel_width", &d.input_channel_width, true}, {"num_hidden_channels", &d.num_hidden_channels, true},
        {"hidden_channel_height", &d.hidden_channel_height, true}, {"hidden_channel_width", &d.hidden_channel_width, true},
        {"num_action_feature_channels", &d.num_action_feature_channels, false}, {"num_blocks", &d.num_blocks, true},
        {"action_size", &d.action_size, true}, {"num_value_hidden_channels", &d.num_value_hidden_channels, true},
        {"discrete_value_size", &d.discrete_value_size, true}};
    d.num_action_feature_channels = 1;
    for (auto& f : ints) {
        const PV* a = attr(*root, f.name);
        if (a && a->kind == PV::Int) { *f.dst = static_cast<int>(a->i); }
        else if (f.required) { return fail(std::string("attribute ") + f.name + " missing"); }
    }
    std::vector<float> w;
    if (!collect(*root, file, zip, prefix, w, err)) { return fail(err); }
    if (desc) { *desc = d; }
    if (weights) { weights->swap(w); }
    return true;
}

} // namespace mz
