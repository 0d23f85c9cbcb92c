// Learner-side sampler of libmzgpu: replaces learner/data_loader.{h,cpp} + the record loaders it samples from
// (ref learner/data_loader.cpp:15-255, learner/pybind.cpp:62-84, environment/base/base_env.h:116-312, the game loaders of go / othello /
// tictactoe / atari).  Same observable behaviour with ONE slave thread (the reference's deterministic contract: thread 0 seeds
// program_seed + 0 and draws [game][position][rotation][...] per sample from one mt19937 through libstdc++'s distributions), own design:
//   * records are parsed once into flat per-game arrays (moves, V / R / L, the P distributions as CSR) instead of tag maps of strings that
//     are re-parsed for every sample;
//   * the two discrete distributions of a sample (game, position) are cached while the priorities do not change — same draws, no O(N) rebuild;
//   * the feature planes — the expensive part: a replay of the game from its first move per sample — are produced on the GPU for the whole
//     batch at once (loader_kernels.hip), directly into the caller's device buffer when it passes one.
#include "config.h"
#include "env.h"
#include "go_dev.h"
#include "loader_dev.h"
#include "net.h"
#include <chrono>
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <numeric>
#include <random>
#include <sstream>

namespace mz {

namespace {

struct LGame {
    int board = 0, seed = 0, d0 = 0, d1 = 0;
    float ret = 0.0f;
    std::vector<int16_t> action;
    std::vector<uint8_t> player;
    std::vector<float> v, r;             // V / R of every move
    std::vector<uint8_t> has_v, has_r, has_l;
    std::vector<int> lives;              // value of the L tag
    std::vector<uint32_t> p_off;         // CSR over moves: entries of the P tag in string order
    std::vector<int> p_action;
    std::vector<float> p_count;
    // Atari: the screens of steps obs_first .. obs_n - 1 (the record keeps the LAST ones) live in device memory, 3 x 96 x 96 bytes each, for as long as
    // the game is in the replay buffer: a sample then costs eight pointers instead of 221 KB of host copies + PCIe (a 1000-step game is 27 MB:
    // 288 GB of HBM hold ten thousand of them)
    int obs_n = 0, obs_first = 0;
    std::shared_ptr<uint8_t> d_obs;
    int size() const { return static_cast<int>(action.size()); }
};

bool parseFloat(const std::string& s, float* out)
{
    try { *out = std::stof(s); } catch (...) { return false; }
    return true;
}
bool parseInt(const std::string& s, int* out)
{
    try { *out = std::stoi(s); } catch (...) { return false; }
    return true;
}

// utils.h:66-91 decompressString
bool hexGunzip(const std::string& hex, std::string* out)
{
    out->clear();
    if (hex.empty()) { return true; }
    if (hex.size() % 2) { return false; }
    std::string bin(hex.size() / 2, '\0');
    auto nib = [](char c) { return c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f' ? c - 'a' + 10 : (c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1)); };
    for (size_t i = 0; i < bin.size(); ++i) {
        const int hi = nib(hex[2 * i]), lo = nib(hex[2 * i + 1]);
        if (hi < 0 || lo < 0) { return false; }
        bin[i] = static_cast<char>(hi * 16 + lo);
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) { return false; }
    zs.next_in = reinterpret_cast<Bytef*>(&bin[0]);
    zs.avail_in = static_cast<uInt>(bin.size());
    std::vector<char> buf(1 << 16);
    int rc = Z_OK;
    while (rc == Z_OK) {
        zs.next_out = reinterpret_cast<Bytef*>(buf.data());
        zs.avail_out = static_cast<uInt>(buf.size());
        rc = inflate(&zs, Z_NO_FLUSH);
        out->append(buf.data(), buf.size() - zs.avail_out);
    }
    inflateEnd(&zs);
    return rc == Z_STREAM_END;
}

float transformValue(float value) // utils.h:93-100
{
    // With <cmath> alone (utils.h:3-12) the unqualified sqrt / fabs of the reference's expression are the C library's DOUBLE functions (libstdc++ puts the
    // float overloads in namespace std only: tests/csrc/overload_check.cpp), so `sign * (sqrt(fabs(v) + 1) - 1) + epsilon * v` is a double expression with
    // one float product (epsilon * v), rounded once by the assignment.  The promotions are spelled out so that no other header can change them.
    const float epsilon = 0.001;
    const float sign_value = (value > 0.0f ? 1.0f : (value == 0.0f ? 0.0f : -1.0f));
    const double r = static_cast<double>(sign_value) * (::sqrt(::fabs(static_cast<double>(value)) + 1) - 1) + static_cast<double>(epsilon * value);
    return static_cast<float>(r);
}

void toDiscreteValue(float value, float* out) // atari.cpp:279-292, out[601]
{
    const int kSize = 601;
    std::fill(out, out + kSize, 0.0f);
    int value_floor = floor(value);
    int value_ceil = ceil(value);
    int shift = kSize / 2;
    int value_floor_shift = std::min(std::max(value_floor + shift, 0), kSize - 1);
    int value_ceil_shift = std::min(std::max(value_ceil + shift, 0), kSize - 1);
    if (value_floor == value_ceil) {
        out[value_floor_shift] = 1.0f;
    } else {
        out[value_floor_shift] = value_ceil - value;
        out[value_ceil_shift] = value - value_floor;
    }
}

} // namespace

class Loader {
public:
    int init(int device, const char* conf);
    int addRecord(const std::string& line);
    void finishLoading() { game_priority_sum_ = std::accumulate(game_priorities_.begin(), game_priorities_.end(), 0.0f); }
    int loadFile(const char* path);
    int sample(float* features, float* action_features, float* policy, float* value, float* reward, float* loss_scale, int* sampled_index, int where);
    bool trace_ = getenv("MZ_TRACE") != nullptr; // prints the host / device split of sample_data when the loader is destroyed
    double trace_ms_[3] = {0, 0, 0};

    int updatePriority(const int* sampled_index, const float* batch_values);
    int numData() const { return num_data_; }
    int numGames() const { return static_cast<int>(games_.size()); }
    int shape(int what) const;
    ~Loader()
    {
        if (trace_ && trace_ms_[2] > 0) {
            fprintf(stderr, "[mz trace] loader: %.3f ms host (draws, targets, staging) + %.3f ms device (uploads, replay, downloads) per batch over %.0f batches\n",
                    trace_ms_[0] / trace_ms_[2], trace_ms_[1] / trace_ms_[2], trace_ms_[2]);
        }
        if (stream_) { (void)hipStreamDestroy(stream_); }
    }

private:
    bool parse(const std::string& content, LGame* g);
    float baseValue(const LGame& g, int pos, bool* ok) const;
    float baseReward(const LGame& g, int pos, bool* ok) const;
    float nStepValue(const LGame& g, int pos, bool* ok) const;
    float priority(const LGame& g, int pos, bool* ok) const;
    void policyOf(const LGame& g, int pos, int rot, float* out) const;
    bool valueOf(const LGame& g, int pos, float* out) const;
    bool rewardOf(const LGame& g, int pos, float* out) const;
    void actionFeaturesOf(const LGame& g, int pos, int rot, float* out);
    int rotateAction(int a, int rot) const { return atari_ ? a : proto_->rot()->fwd[rot][a]; }
    int sampleIndex(std::discrete_distribution<>& dis) { return dis(gen_); }
    int randInt() { return int_dist_(gen_); }
    int ensureDevice(int B);

    WorkerConfig cfg_;
    int device_ = 0;
    hipStream_t stream_ = nullptr;
    bool atari_ = false, muzero_ = false;
    bool device_failure_ = false; // parse(): the record failed for lack of device memory, not for its content
    int A_ = 0, P_ = 0, feat_size_ = 0, act_feat_size_ = 0, value_size_ = 1, max_len_ = 0;
    std::unique_ptr<GameEnv> proto_; // a fresh environment of the configured game: rotation tables, root snapshot, shapes
    // ReplayBuffer (data_loader.cpp:15-82)
    int num_data_ = 0;
    float game_priority_sum_ = 0.0f;
    std::deque<float> game_priorities_;
    std::deque<std::deque<float>> position_priorities_;
    std::deque<LGame> games_;
    // cached distributions (valid while the weights they were built from are unchanged)
    bool game_dis_valid_ = false;
    std::discrete_distribution<> game_dis_;
    std::deque<std::unique_ptr<std::discrete_distribution<>>> pos_dis_;
    std::mt19937 gen_;
    std::uniform_int_distribution<int> int_dist_;
    // device side
    GoDevice godev_;
    int dev_batch_ = 0, slots_ = 0;
    DevBuf<int> d_path_, d_int_;        // identity path / hslot; per batch: path_len, path_action, pos
    DevBuf<uint8_t> d_rot_, d_meta_;
    DevBuf<float> d_feat_;
    PinBuf<int> h_int_;
    PinBuf<uint8_t> h_rot_, h_meta_;
    PoolView pv_{};
};

int Loader::shape(int what) const
{
    const int U = cfg_.learner_muzero_unrolling_step;
    switch (what) {
        case 0: return cfg_.learner_batch_size;
        case 1: return feat_size_;                                   // features per sample
        case 2: return muzero_ ? U * act_feat_size_ : 0;             // action features per sample
        case 3: return muzero_ ? (U + 1) * A_ : A_;                  // policy per sample
        case 4: return muzero_ ? (U + 1) * value_size_ : value_size_; // value per sample
        case 5: return muzero_ ? U * value_size_ : 0;                // reward per sample
        default: return MZ_ERR_ARG;
    }
}

int Loader::init(int device, const char* conf)
{
    if (!conf || !cfg_.loadFromString(conf)) { return MZ_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        setError("mz_loader_create: no such GPU %d (%d visible) — libmzgpu has no CPU path", device, ndev);
        return MZ_ERR_DEVICE;
    }
    device_ = device;
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    if (cfg_.nn_type_name != "alphazero" && cfg_.nn_type_name != "muzero") { setError("nn_type_name must be alphazero or muzero (ref data_loader.cpp:135-141)"); return MZ_ERR_ARG; }
    muzero_ = cfg_.nn_type_name == "muzero";
    atari_ = cfg_.env_game == "atari";
    proto_ = createGameEnv(cfg_.env_game, cfg_.env_board_size, cfg_.env_go_komi, cfg_.env_atari_name, cfg_.env_atari_episode_length, cfg_.env_go_ko_rule, 1);
    if (!proto_) { return MZ_ERR_ARG; }
    if (!atari_ && !proto_->hasDeviceTwin()) { setError("loader: no device engine for %s at this board size", proto_->name().c_str()); return MZ_ERR_ARG; }
    A_ = proto_->policySize();
    P_ = atari_ ? 36 : proto_->boardSize() * proto_->boardSize();
    feat_size_ = proto_->featureSize();
    act_feat_size_ = atari_ ? 18 * 36 : P_;
    value_size_ = atari_ ? 601 : 1;
    const int seed = cfg_.program_auto_seed ? static_cast<int>(std::random_device()()) : cfg_.program_seed + 0; // DataLoaderThread::initialize, id 0
    gen_.seed(seed);
    return MZ_OK;
}

// the record state machine of base_env.h:150-205, writing the flat game
bool Loader::parse(const std::string& content, LGame* g)
{
    std::string key, value;
    int state = '(';
    bool accept_move = false, escape_next = false;
    int board_size = cfg_.env_board_size > 0 ? cfg_.env_board_size : (atari_ ? 0 : proto_->boardSize());
    std::vector<std::pair<std::string, std::string>> tags; // only a few are read: linear search keeps the reference's "last assignment wins"
    auto setTag = [&](std::vector<std::pair<std::string, std::string>>& m, const std::string& k, std::string&& v) {
        for (auto& kv : m) { if (kv.first == k) { kv.second = std::move(v); return; } }
        m.emplace_back(k, std::move(v));
    };
    std::vector<std::vector<std::pair<std::string, std::string>>> infos;
    for (char c : content) {
        switch (state) {
            case '(':
                if (!accept_move) { accept_move = (c == '('); }
                else { state = (c == ';') ? c : 'x'; accept_move = false; }
                break;
            case ';':
                if (c == ';') { accept_move = true; }
                else if (c == '[' || c == ')') { state = c; }
                else if (std::isgraph(c)) { key += c; }
                break;
            case '[':
                if (c == '\\' && !escape_next) {
                    escape_next = true;
                } else if (c != ']' || escape_next) {
                    value += c;
                    escape_next = false;
                } else {
                    if (accept_move) {
                        int action_id = 0;
                        if (value.size() && std::isdigit(value[0])) { if (!parseInt(value, &action_id)) { return false; } }
                        else if (value.size() != 2) { action_id = board_size * board_size; } // sgfStringToActionID (sgf_loader.cpp:123-129)
                        else { action_id = ((board_size - 1) - (std::toupper(value[1]) - 'A')) * board_size + (std::toupper(value[0]) - 'A'); }
                        g->action.push_back(static_cast<int16_t>(action_id));
                        g->player.push_back(key.empty() ? 0 : (key[0] == 'B' || key[0] == 'b' ? 1 : (key[0] == 'W' || key[0] == 'w' ? 2 : 0)));
                        infos.emplace_back();
                        accept_move = false;
                    } else if (!infos.empty()) {
                        setTag(infos.back(), key, std::move(value));
                    } else {
                        if (key == "SZ") { if (!parseInt(value, &board_size)) { return false; } }
                        setTag(tags, key, std::move(value));
                    }
                    key.clear();
                    value.clear();
                    state = ';';
                }
                break;
            case ')':
                break;
        }
    }
    if (state != ')') { return false; }
    auto tag = [&](const char* k) -> const std::string* { for (auto& kv : tags) { if (kv.first == k) { return &kv.second; } } return nullptr; };
    g->board = board_size;
    const int n = g->size();
    if (!atari_ && board_size != proto_->boardSize()) { setError("loader: record of a %dx%d board, the loader is configured for %dx%d", board_size, board_size, proto_->boardSize(), proto_->boardSize()); return false; }
    for (int i = 0; i < n; ++i) { if (g->action[i] < 0 || g->action[i] >= A_) { setError("loader: action %d out of range in a record", g->action[i]); return false; } }
    // The device replay (loader_kernels.hip) takes the mover from the move's parity: the reference replays act(action) with the RECORDED player
    // (base_env.h:235-241), so a record whose colours do not alternate from the first player (handicap stones, hand-edited files) would give other
    // planes there.  Self-play records always alternate; anything else is refused by name instead of being replayed differently.
    if (!atari_) {
        for (int i = 0; i < n; ++i) {
            if (g->player[i] != 1 + (i & 1)) { setError("loader: move %d of a record is played by %s, the device replay needs alternating colours from B", i, g->player[i] == 1 ? "B" : g->player[i] == 2 ? "W" : "nobody"); return false; }
        }
    }
    // getReturn() = stof(RE) (base_env.h:300); only read for board games, but every record carries it
    if (const std::string* re = tag("RE")) { if (!parseFloat(*re, &g->ret)) { g->ret = 0.0f; } }
    if (const std::string* sd = tag("SD")) { (void)parseInt(*sd, &g->seed); }
    // getDataRange (base_env.h:267-276)
    const std::string* dlen = tag("DLEN");
    if (!dlen || dlen->empty()) { g->d0 = 0; g->d1 = std::max(0, n - 1); }
    else {
        if (!parseInt(*dlen, &g->d0) || dlen->find('-') == std::string::npos || !parseInt(dlen->substr(dlen->find('-') + 1), &g->d1)) { setError("loader: bad DLEN tag"); return false; }
    }
    if (g->d0 < 0 || g->d1 < g->d0) { setError("loader: bad data range %d-%d", g->d0, g->d1); return false; }
    g->v.assign(n, 0.0f); g->r.assign(n, 0.0f); g->has_v.assign(n, 0); g->has_r.assign(n, 0); g->has_l.assign(n, 0); g->lives.assign(n, 0);
    g->p_off.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        for (auto& kv : infos[i]) {
            if (kv.first == "V") { g->has_v[i] = parseFloat(kv.second, &g->v[i]); }
            else if (kv.first == "R") { g->has_r[i] = parseFloat(kv.second, &g->r[i]); }
            else if (kv.first == "L") { g->has_l[i] = 1; (void)parseInt(kv.second, &g->lives[i]); }
            else if (kv.first == "P" && !kv.second.empty()) { // base_env.h:250-260
                std::string tmp;
                std::istringstream iss(kv.second);
                while (std::getline(iss, tmp, ',')) {
                    int a = 0;
                    float count = 0;
                    if (tmp.find(':') == std::string::npos || !parseInt(tmp.substr(0, tmp.find(':')), &a) || !parseFloat(tmp.substr(tmp.find(':') + 1), &count) || a < 0 || a >= A_) {
                        setError("loader: bad P tag entry '%s'", tmp.c_str());
                        return false;
                    }
                    g->p_action.push_back(a);
                    g->p_count.push_back(count);
                }
            }
        }
        g->p_off[i + 1] = static_cast<uint32_t>(g->p_action.size());
    }
    if (atari_) { // atari.cpp:179-184,237-249: observations aligned to the END of the game
        g->obs_n = n + 1;
        g->obs_first = n + 1;
        const std::string* obs = tag("OBS");
        if (obs && !obs->empty()) {
            std::string raw;
            const size_t frame = size_t(3) * 96 * 96;
            if (!hexGunzip(*obs, &raw) || raw.size() % frame) { setError("loader: the OBS tag does not hold whole 3x96x96 screens"); return false; }
            const size_t kept = std::min(raw.size() / frame, size_t(n) + 1);
            if (kept > 0) {
                uint8_t* d = nullptr;
                if (hipSetDevice(device_) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d), kept * frame) != hipSuccess) { device_failure_ = true; setError("loader: no device memory for the observations of a game (%zu bytes)", kept * frame); return false; }
                g->d_obs = std::shared_ptr<uint8_t>(d, [](uint8_t* q) { (void)hipFree(q); });
                if (hipMemcpy(d, raw.data() + raw.size() - kept * frame, kept * frame, hipMemcpyHostToDevice) != hipSuccess) { device_failure_ = true; setError("loader: upload of the observations failed"); return false; }
                g->obs_first = n + 1 - static_cast<int>(kept);
            }
        }
    }
    return true;
}

float Loader::baseValue(const LGame& g, int pos, bool* ok) const
{
    if (pos >= g.size()) { return 0.0f; }
    if (!g.has_v[pos]) { *ok = false; }
    return g.v[pos];
}
float Loader::baseReward(const LGame& g, int pos, bool* ok) const
{
    if (pos >= g.size()) { return 0.0f; }
    if (!g.has_r[pos]) { *ok = false; }
    return g.r[pos];
}

float Loader::nStepValue(const LGame& g, int pos, bool* ok) const // atari.cpp:259-277
{
    const int n_step = cfg_.learner_n_step_return;
    const float discount = cfg_.actor_mcts_reward_discount;
    const size_t size = static_cast<size_t>(g.size());
    size_t bootstrap_index = pos + n_step;
    float value = 0.0f;
    float n_step_value = ((bootstrap_index < size && !g.has_l[bootstrap_index]) ? std::pow(discount, n_step) * baseValue(g, static_cast<int>(bootstrap_index), ok) : 0.0f);
    for (size_t index = pos; index < std::min(bootstrap_index, size); ++index) {
        if (g.has_l[index] && g.lives[index] > 0) { return value; }
        float reward = baseReward(g, static_cast<int>(index), ok);
        value += std::pow(discount, index - pos) * reward;
    }
    value += n_step_value;
    return value;
}

float Loader::priority(const LGame& g, int pos, bool* ok) const
{
    if (!atari_) { return 1.0f; }
    return fabs(nStepValue(g, pos, ok) - baseValue(g, pos, ok)) + 1e-6; // atari.h:117
}

void Loader::policyOf(const LGame& g, int pos, int rot, float* out) const // base_env.h:243-265
{
    if (pos >= g.size()) { std::fill(out, out + A_, 1.0f / A_); return; }
    std::fill(out, out + A_, 0.0f);
    const uint32_t b = g.p_off[pos], e = g.p_off[pos + 1];
    if (b == e) { out[rotateAction(g.action[pos], rot)] = 1.0f; return; }
    float total = 0.0f;
    for (uint32_t k = b; k < e; ++k) {
        out[rotateAction(g.p_action[k], rot)] = g.p_count[k];
        total += g.p_count[k];
    }
    for (int a = 0; a < A_; ++a) { out[a] /= total; }
}

bool Loader::valueOf(const LGame& g, int pos, float* out) const
{
    if (!atari_) { out[0] = g.ret; return true; } // go.h:137, othello.h:79, tictactoe.h:47
    bool ok = true;
    toDiscreteValue(pos < g.size() ? transformValue(nStepValue(g, pos, &ok)) : 0.0f, out); // atari.h:115
    return ok;
}
bool Loader::rewardOf(const LGame& g, int pos, float* out) const
{
    bool ok = true;
    const float r = baseReward(g, pos, &ok);
    if (!atari_) { out[0] = r; return ok; }
    toDiscreteValue(pos < g.size() ? transformValue(r) : 0.0f, out); // atari.h:116
    return ok;
}

void Loader::actionFeaturesOf(const LGame& g, int pos, int rot, float* out)
{
    std::fill(out, out + act_feat_size_, 0.0f);
    const int size = g.size();
    if (atari_) { // atari.cpp:223-235
        const int a = pos < size ? g.action[pos] : randInt() % 18;
        std::fill(out + a * 36, out + (a + 1) * 36, 1.0f);
    } else if (cfg_.env_game == "tictactoe") { // tictactoe.cpp:148-155
        out[pos < size ? rotateAction(g.action[pos], rot) : randInt() % P_] = 1.0f;
    } else if (pos < size) { // go.cpp:725-737, othello.cpp:264-276
        if (g.action[pos] != P_) { out[rotateAction(g.action[pos], rot)] = 1.0f; }
    } else {
        // the reference's index can be P_ (one past its private P_-wide vector, go.cpp:733-734) when the game has more than P_ moves: the draw
        // is kept, the stray write is not — every value the reference hands on is unchanged
        const int a = randInt() % (P_ + 1);
        if (a < size && a < P_) { out[a] = 1.0f; }
    }
}

int Loader::addRecord(const std::string& line_in)
{
    // a line of the server's sgf file is the bare record; `SelfPlay <terminal> <len> <len> <return> <record> #` lines are accepted too
    std::string content = line_in;
    if (content.rfind("SelfPlay ", 0) == 0) {
        size_t p = 0;
        for (int k = 0; k < 5 && p != std::string::npos; ++k) { p = content.find(' ', p + (k ? 1 : 0)); }
        if (p == std::string::npos) { setError("loader: malformed SelfPlay line"); return MZ_ERR_ARG; }
        content = content.substr(p + 1);
        const size_t e = content.rfind(" #");
        if (e != std::string::npos) { content = content.substr(0, e); }
    }
    LGame g;
    device_failure_ = false;
    if (!parse(content, &g)) { return device_failure_ ? MZ_ERR_DEVICE : 0; } // like DataLoaderThread::addEnvironmentLoader: a record that does not load is skipped — but a record the DEVICE had no room for is an error, not a silent loss of training data
    if (g.d1 >= g.size() + 1 && g.size() > 0) { setError("loader: data range beyond the game"); return 0; }
    // ReplayBuffer::addData (data_loader.cpp:24-50)
    std::deque<float> position_priorities(g.d1 + 1, 0.0f);
    float game_priority = 0.0f;
    bool ok = true;
    for (int i = g.d0; i <= g.d1; ++i) {
        position_priorities[i] = std::pow((cfg_.learner_use_per ? priority(g, i, &ok) : 1.0f), cfg_.learner_per_alpha);
        game_priority += position_priorities[i];
    }
    if (!ok) { setError("loader: a record without V / R tags cannot be prioritised"); return 0; }
    num_data_ += (g.d1 - g.d0 + 1);
    max_len_ = std::max(max_len_, g.size());
    position_priorities_.push_back(std::move(position_priorities));
    game_priorities_.push_back(game_priority);
    games_.push_back(std::move(g));
    pos_dis_.emplace_back();
    const size_t replay_buffer_max_size = static_cast<size_t>(cfg_.zero_replay_buffer * cfg_.zero_num_games_per_iteration);
    while (position_priorities_.size() > replay_buffer_max_size) {
        num_data_ -= (games_.front().d1 - games_.front().d0 + 1);
        position_priorities_.pop_front();
        game_priorities_.pop_front();
        games_.pop_front();
        pos_dis_.pop_front();
    }
    game_dis_valid_ = false;
    return 1;
}

int Loader::loadFile(const char* path)
{
    std::ifstream fin(path, std::ifstream::in);
    if (!fin) { setError("loader: cannot open %s", path); return MZ_ERR_ARG; }
    int loaded = 0;
    for (std::string content; std::getline(fin, content);) {
        const int rc = addRecord(content);
        if (rc < 0) { return rc; }
        loaded += rc;
    }
    finishLoading();
    return loaded;
}

int Loader::ensureDevice(int B)
{
    const int slots = std::max(max_len_, 1) + 2;
    if (B <= dev_batch_ && slots <= slots_) { return MZ_OK; }
    MZ_HIP(hipSetDevice(device_));
    MZ_HIP(hipStreamSynchronize(stream_));
    dev_batch_ = std::max(B, dev_batch_);
    slots_ = std::max(slots, slots_);
    if (!d_feat_.alloc(size_t(dev_batch_) * feat_size_) || !d_rot_.alloc(dev_batch_) || !h_rot_.alloc(dev_batch_)) { setError("loader: allocation failed"); return MZ_ERR_DEVICE; }
    if (atari_) {
        if (!d_meta_.alloc(size_t(dev_batch_) * kAtariMetaBytes) || !h_meta_.alloc(size_t(dev_batch_) * kAtariMetaBytes)) { setError("loader: allocation failed (observation descriptors)"); return MZ_ERR_DEVICE; }
        return MZ_OK;
    }
    const int MD = slots_ + 1;
    const GameEnv& e = *proto_;
    const int* inv[8];
    const int* fwd[8];
    for (int r = 0; r < 8; ++r) { inv[r] = e.rot()->inv[r].data(); fwd[r] = e.rot()->fwd[r].data(); }
    int rc = godev_.init(device_, dev_batch_, e.boardSize(), cfg_.env_go_komi, A_, slots_, MD, stream_, inv, fwd, e.zobristKeys(), e.deviceKind(), e.turnKey());
    if (rc) { return rc; }
    for (int b = 0; b < dev_batch_; ++b) { e.exportDeviceRoot(godev_.hostSnap(b)); }
    if ((rc = godev_.uploadRoots())) { return rc; }
    // a sample's "tree" is a chain: node d = the position after d moves, kept in slot d
    const size_t BM = size_t(dev_batch_) * MD;
    if (!d_path_.alloc(2 * BM) || !d_int_.alloc(size_t(dev_batch_) * 2 + BM) || !h_int_.alloc(size_t(dev_batch_) * 2 + BM)) { setError("loader: allocation failed (paths)"); return MZ_ERR_DEVICE; }
    std::vector<int> ident(2 * BM);
    for (int b = 0; b < dev_batch_; ++b) { for (int d = 0; d < MD; ++d) { ident[size_t(b) * MD + d] = d; ident[BM + size_t(b) * MD + d] = d; } }
    MZ_HIP(hipMemcpy(d_path_.p, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
    pv_ = PoolView{};
    pv_.games = dev_batch_; pv_.cap = MD; pv_.A = A_; pv_.max_depth = MD;
    pv_.path = d_path_.p; pv_.hslot = d_path_.p + BM;
    pv_.path_len = d_int_.p; pv_.path_action = d_int_.p + 2 * size_t(dev_batch_);
    return MZ_OK;
}

int Loader::sample(float* features, float* action_features, float* policy, float* value, float* reward, float* loss_scale, int* sampled_index, int where)
{
    const int B = cfg_.learner_batch_size, U = cfg_.learner_muzero_unrolling_step;
    if (games_.empty()) { setError("sample_data: the replay buffer is empty"); return MZ_ERR_STATE; }
    if (!features || !policy || !value || !loss_scale || !sampled_index || (muzero_ && (!action_features || !reward))) { setError("sample_data: NULL buffer"); return MZ_ERR_ARG; }
    if (B > kRotPackGames) { setError("sample_data: learner_batch_size %d > %d", B, kRotPackGames); return MZ_ERR_ARG; }
    int rc = ensureDevice(B);
    if (rc) { return rc; }
    MZ_HIP(hipSetDevice(device_));
    const auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = tnow();
    const int MD = slots_ + 1;
    const size_t np = size_t(shape(3)), nv = size_t(shape(4)), nr = size_t(shape(5)), na = size_t(shape(2));
    std::vector<float> h_policy(B * np), h_value(B * nv), h_reward(B * std::max<size_t>(nr, 1)), h_af(B * std::max<size_t>(na, 1)), h_ls(B);
    std::vector<int> h_si(2 * size_t(B));
    if (!game_dis_valid_) { game_dis_ = std::discrete_distribution<>(game_priorities_.begin(), game_priorities_.end()); game_dis_valid_ = true; }
    for (int b = 0; b < B; ++b) {
        // ReplayBuffer::sampleEnvAndPos + the rotation draw (data_loader.cpp:52-63,148 / 166)
        const int env_id = sampleIndex(game_dis_);
        if (!pos_dis_[env_id]) { pos_dis_[env_id] = std::make_unique<std::discrete_distribution<>>(position_priorities_[env_id].begin(), position_priorities_[env_id].end()); }
        const int pos = sampleIndex(*pos_dis_[env_id]);
        const int rot = randInt() % 8;
        const LGame& g = games_[env_id];
        float ls = 1.0f; // ReplayBuffer::getLossScale (data_loader.cpp:73-82)
        if (cfg_.learner_use_per) {
            const float prob = position_priorities_[env_id][pos] / game_priority_sum_;
            ls = std::pow((num_data_ * prob), (-cfg_.learner_per_init_beta));
        }
        h_ls[b] = ls;
        h_si[2 * b] = env_id;
        h_si[2 * b + 1] = pos;
        // ---- features: staged for the device ----
        h_rot_.p[b] = static_cast<uint8_t>(atari_ ? 0 : rot);
        if (atari_) { // atari.cpp:199-221: per sample eight (screen pointer, action value, valid) triples
            uint8_t* m = h_meta_.p + size_t(b) * kAtariMetaBytes;
            const size_t frame = size_t(3) * 96 * 96;
            float av[8];
            uint64_t ptr[8];
            for (int k = 0; k < 8; ++k) {
                const int i = pos - 7 + k;
                const int action_id = (i - 1 < 0 ? 0 : (i - 1 >= g.size() ? randInt() % 18 : g.action[i - 1]));
                av[k] = action_id * 1.0f / 18;
                uint8_t valid = 0;
                ptr[k] = 0;
                if (i >= 0) {
                    const int idx = i < g.obs_n ? i : g.obs_n - 1;
                    if (idx < g.obs_first || !g.d_obs) { setError("sample_data: the record of game %d keeps no observation for step %d (replay of the Atari environment is not available)", env_id, i); return MZ_ERR_STATE; }
                    ptr[k] = reinterpret_cast<uint64_t>(g.d_obs.get() + size_t(idx - g.obs_first) * frame);
                    valid = 1;
                }
                m[96 + k] = valid;
            }
            memcpy(m, ptr, sizeof(ptr));
            memcpy(m + 64, av, sizeof(av));
        } else {
            int* hi = h_int_.p;
            hi[size_t(dev_batch_) + b] = std::min(pos, g.size());                 // moves to replay
            int* pact = hi + 2 * size_t(dev_batch_) + size_t(b) * MD;
            pact[0] = -1;
            for (int d = 1; d <= std::min(pos, g.size()); ++d) { pact[d] = g.action[d - 1]; }
        }
        // ---- targets (host) ----
        bool ok = true;
        if (!muzero_) {
            policyOf(g, pos, rot, h_policy.data() + b * np);
            ok &= valueOf(g, pos, h_value.data() + b * nv);
        } else {
            for (int step = 0; step <= U; ++step) {
                if (step < U) { actionFeaturesOf(g, pos + step, rot, h_af.data() + b * na + size_t(step) * act_feat_size_); }
                policyOf(g, pos + step, rot, h_policy.data() + b * np + size_t(step) * A_);
                ok &= valueOf(g, pos + step, h_value.data() + b * nv + size_t(step) * value_size_);
                if (step < U) { ok &= rewardOf(g, pos + step, h_reward.data() + b * nr + size_t(step) * value_size_); }
            }
        }
        if (!ok) { setError("sample_data: game %d lacks the V / R tags its targets need", env_id); return MZ_ERR_STATE; }
    }
    // ---- features on the device ----
    const double t_host = tnow();
    float* d_out = where == MZ_DEVICE ? features : d_feat_.p;
    MZ_HIP(hipMemcpyAsync(d_rot_.p, h_rot_.p, B, hipMemcpyHostToDevice, stream_));
    if (atari_) {
        MZ_HIP(hipMemcpyAsync(d_meta_.p, h_meta_.p, size_t(B) * kAtariMetaBytes, hipMemcpyHostToDevice, stream_));
        if ((rc = loaderExpandAtari(d_meta_.p, B, d_out, stream_))) { return rc; }
    } else {
        MZ_HIP(hipMemcpyAsync(d_int_.p + size_t(dev_batch_), h_int_.p + size_t(dev_batch_), (size_t(dev_batch_) + size_t(dev_batch_) * MD) * sizeof(int), hipMemcpyHostToDevice, stream_));
        if ((rc = loaderReplayFeatures(godev_, pv_, B, d_int_.p + size_t(dev_batch_), d_rot_.p, d_out, stream_))) { return rc; }
    }
    auto put = [&](void* dst, const void* src, size_t bytes) -> int {
        if (bytes == 0) { return MZ_OK; }
        if (where == MZ_DEVICE) { MZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream_)); }
        else { memcpy(dst, src, bytes); }
        return MZ_OK;
    };
    if (where != MZ_DEVICE) { MZ_HIP(hipMemcpyAsync(features, d_feat_.p, size_t(B) * feat_size_ * sizeof(float), hipMemcpyDeviceToHost, stream_)); }
    if ((rc = put(policy, h_policy.data(), B * np * sizeof(float))) || (rc = put(value, h_value.data(), B * nv * sizeof(float))) ||
        (rc = put(loss_scale, h_ls.data(), B * sizeof(float))) || (rc = put(sampled_index, h_si.data(), 2 * size_t(B) * sizeof(int)))) { return rc; }
    if (muzero_ && ((rc = put(action_features, h_af.data(), B * na * sizeof(float))) || (rc = put(reward, h_reward.data(), B * nr * sizeof(float))))) { return rc; }
    MZ_HIP(hipStreamSynchronize(stream_));
    if (trace_) { const double t_end = tnow(); trace_ms_[0] += t_host - t_start; trace_ms_[1] += t_end - t_host; trace_ms_[2] += 1; }
    return MZ_OK;
}

int Loader::updatePriority(const int* sampled_index, const float* batch_values) // data_loader.cpp:233-253
{
    const int B = cfg_.learner_batch_size, U = cfg_.learner_muzero_unrolling_step;
    for (int b = 0; b < B; ++b) {
        const int env_id = sampled_index[2 * b], pos_id = sampled_index[2 * b + 1];
        if (env_id < 0 || env_id >= numGames() || pos_id < 0 || pos_id >= static_cast<int>(position_priorities_[env_id].size())) { setError("update_priority: index out of range"); return MZ_ERR_ARG; }
        LGame& g = games_[env_id];
        for (int step = 0; step <= U; ++step) {
            const float new_value = invertValueHost(batch_values[size_t(step) * B + b]);
            // setActionPairInfo(pos, "V", std::to_string(v)): the value goes through its 6-decimal text form
            if (pos_id + step < g.size()) { g.has_v[pos_id + step] = parseFloat(std::to_string(new_value), &g.v[pos_id + step]); }
        }
        bool ok = true;
        position_priorities_[env_id][pos_id] = std::pow(priority(g, pos_id, &ok), cfg_.learner_per_alpha);
        pos_dis_[env_id].reset();
    }
    for (size_t i = 0; i < game_priorities_.size(); ++i) { game_priorities_[i] = std::accumulate(position_priorities_[i].begin(), position_priorities_[i].end(), 0.0f); }
    game_priority_sum_ = std::accumulate(game_priorities_.begin(), game_priorities_.end(), 0.0f);
    game_dis_valid_ = false;
    return MZ_OK;
}

} // namespace mz

struct mz_loader { mz::Loader l; };

extern "C" {

mz_loader* mz_loader_create(int device, const char* conf)
{
    std::unique_ptr<mz_loader> l(new mz_loader());
    if (l->l.init(device, conf) != MZ_OK) { return nullptr; }
    return l.release();
}
void mz_loader_destroy(mz_loader* l) { delete l; }
int mz_loader_add_record(mz_loader* l, const char* line)
{
    if (!l || !line) { mz::setError("mz_loader_add_record: NULL argument"); return MZ_ERR_ARG; }
    const int rc = l->l.addRecord(line);
    l->l.finishLoading();
    return rc;
}
int mz_loader_load_data_from_file(mz_loader* l, const char* path)
{
    if (!l || !path) { mz::setError("mz_loader_load_data_from_file: NULL argument"); return MZ_ERR_ARG; }
    return l->l.loadFile(path);
}
int mz_loader_sample_data(mz_loader* l, float* features, float* action_features, float* policy, float* value, float* reward, float* loss_scale,
                          int* sampled_index, int where)
{
    if (!l) { mz::setError("NULL loader"); return MZ_ERR_ARG; }
    return l->l.sample(features, action_features, policy, value, reward, loss_scale, sampled_index, where);
}
int mz_loader_update_priority(mz_loader* l, const int* sampled_index, const float* batch_values)
{
    if (!l || !sampled_index || !batch_values) { mz::setError("mz_loader_update_priority: NULL argument"); return MZ_ERR_ARG; }
    return l->l.updatePriority(sampled_index, batch_values);
}
float mz_transform_value(float v) { return mz::transformValue(v); }
int mz_loader_num_data(const mz_loader* l) { return l ? l->l.numData() : MZ_ERR_ARG; }
int mz_loader_num_games(const mz_loader* l) { return l ? l->l.numGames() : MZ_ERR_ARG; }
int mz_loader_shape(const mz_loader* l, int what) { return l ? l->l.shape(what) : MZ_ERR_ARG; }

} // extern "C"
