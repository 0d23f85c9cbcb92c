// Host rules engines of libmzgpu — see env.h.  Not a translation of the reference's data structures:
//   Go      : flat byte board + on-demand early-exit flood fills for captures (no incremental blocks /
//             liberty bitsets / areas / Benson), one whole-board group pass only when a legal mask is
//             needed, positional-superko set as a small open-addressing table that is copied with the env,
//             8-deep ring of stone bitboards for the history planes.
//   Othello : two 64-bit bitboards, shift-and-mask move generation (8x8 and smaller).
//   TicTacToe: two 9-bit masks.
// All feature planes are written straight into caller memory (the worker's pinned staging buffer).
#include "env.h"
#include "go_dev.h"
#include "common.h"
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>

namespace mz {

// ---------------------------------------------------------------------------------------------
// rotation tables (ref utils/rotation.h:21-29,51-93: float centre arithmetic with truncation)
// ---------------------------------------------------------------------------------------------
static int rotatePos(int rotation, int pos, int n)
{
    if (pos == n * n) { return pos; }
    const float center = (n - 1) / 2.0;
    const float x = pos % n - center, y = pos / n - center;
    float rx = x, ry = y;
    switch (rotation) {
        case 0: rx = x; ry = y; break;
        case 1: rx = y; ry = -x; break;
        case 2: rx = -x; ry = -y; break;
        case 3: rx = -y; ry = x; break;
        case 4: rx = x; ry = -y; break;
        case 5: rx = -y; ry = -x; break;
        case 6: rx = -x; ry = y; break;
        case 7: rx = y; ry = x; break;
    }
    return static_cast<int>((ry + center) * n + (rx + center));
}

void RotationTables::build(int n, int num_actions)
{
    static const int reversed[8] = {0, 3, 2, 1, 4, 5, 6, 7};
    board_size = n;
    for (int r = 0; r < 8; ++r) {
        fwd[r].resize(num_actions);
        inv[r].resize(n * n);
        for (int a = 0; a < num_actions; ++a) { fwd[r][a] = (a >= n * n) ? a : rotatePos(r, a, n); }
        for (int p = 0; p < n * n; ++p) { inv[r][p] = rotatePos(reversed[r], p, n); }
    }
}

static const RotationTables* rotationTables(int n, int num_actions)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::unique_ptr<RotationTables>> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto& slot = cache[{n, num_actions}];
    if (!slot) {
        slot = std::make_unique<RotationTables>();
        slot->build(n, num_actions);
    }
    return slot.get();
}

void GameEnv::featureBits(int rotation, uint32_t* out) const
{
    const int P = boardSize() * boardSize(), W32 = (P + 31) / 32, C = numInputChannels();
    std::vector<float> f(size_t(C) * P);
    features(rotation, f.data());
    for (int c = 0; c < C; ++c) {
        for (int w = 0; w < W32; ++w) { out[c * W32 + w] = 0; }
        for (int p = 0; p < P; ++p) { if (f[size_t(c) * P + p] != 0.0f) { out[c * W32 + (p >> 5)] |= 1u << (p & 31); } }
    }
}

static inline float scoreOf(int winner) { return winner == 1 ? 1.0f : (winner == 2 ? -1.0f : 0.0f); }

// ---------------------------------------------------------------------------------------------
// TicTacToe (ref tictactoe.cpp:11-146)
// ---------------------------------------------------------------------------------------------
// ref utils/sgf_loader.cpp:89-99 boardCoordinateStringToActionID (the conversion behind BaseBoardAction(action_string_args), base_env.h:326-333)
int GameEnv::actionFromString(const std::string& str) const
{
    const int n = boardSize();
    std::string up = str;
    for (char& c : up) { c = static_cast<char>(std::toupper(static_cast<unsigned char>(c))); }
    if (up == "PASS") { return n * n; }
    if (str.size() < 2) { return -1; }
    const int x = up[0] - 'A' + (up[0] > 'I' ? -1 : 0);
    const int y = atoi(str.substr(1).c_str()) - 1;
    return y * n + x;
}

class TicTacToe final : public GameEnv {
public:
    TicTacToe() { rot_ = rotationTables(3, 9); reset(); }
    std::unique_ptr<GameEnv> clone() const override { return std::make_unique<TicTacToe>(*this); }
    void copyFrom(const GameEnv& o) override { *this = static_cast<const TicTacToe&>(o); }
    void reset() override { turn_ = 1; action_ids_.clear(); action_players_.clear(); m_[0] = m_[1] = 0; }
    bool isLegal(int a, int) const override { return a >= 0 && a < 9 && !((m_[0] | m_[1]) >> a & 1); }
    bool act(int a, int player) override
    {
        if (!isLegal(a, player)) { return false; }
        actUnchecked(a, player);
        return true;
    }
    void actUnchecked(int a, int player) override
    {
        m_[player - 1] |= 1u << a;
        action_ids_.push_back(static_cast<int16_t>(a));
        action_players_.push_back(static_cast<uint8_t>(player));
        turn_ = 3 - player;
    }
    void legalMask(uint8_t* out) const override { for (int a = 0; a < 9; ++a) { out[a] = isLegal(a, turn_); } }
    int winner() const
    {
        static const unsigned lines[8] = {0007, 0070, 0700, 0111, 0222, 0444, 0421, 0124};
        for (unsigned l : lines) {
            if ((m_[0] & l) == l) { return 1; }
            if ((m_[1] & l) == l) { return 2; }
        }
        return 0;
    }
    bool isTerminal() const override { return winner() != 0 || (m_[0] | m_[1]) == 0777; }
    float evalScore(bool is_resign) const override { return scoreOf(is_resign ? 3 - turn_ : winner()); }
    void features(int r, float* out) const override
    {
        const unsigned own = m_[turn_ - 1], opp = m_[2 - turn_];
        const int* map = rot_->inv[r].data();
        for (int p = 0; p < 9; ++p) {
            out[p] = (own >> map[p]) & 1 ? 1.0f : 0.0f;
            out[9 + p] = (opp >> map[p]) & 1 ? 1.0f : 0.0f;
            out[18 + p] = turn_ == 1 ? 1.0f : 0.0f;
            out[27 + p] = turn_ == 2 ? 1.0f : 0.0f;
        }
    }
    bool hasDeviceTwin() const override { return true; }
    int deviceKind() const override { return 2; }
    void exportDeviceRoot(void* dst) const override // the fields of GoRootSnapshot the TicTacToe device engine reads (go_body.h tttLeafBody)
    {
        GoRootSnapshot& s = *static_cast<GoRootSnapshot*>(dst);
        s.stones[0][0] = m_[0];
        s.stones[1][0] = m_[1];
        s.hash = 0;
        s.hist_len = 0;
        s.turn = turn_;
        s.nmoves = static_cast<int32_t>(action_ids_.size());
        s.passes = 0;
    }
    int numInputChannels() const override { return 4; }
    int boardSize() const override { return 3; }
    int policySize() const override { return 9; }
    std::string name() const override { return "tictactoe"; }
    std::vector<std::pair<std::string, std::string>> loaderTags() const override { return {{"SZ", "3"}}; }

private:
    unsigned m_[2];
};

// ---------------------------------------------------------------------------------------------
// Othello, board <= 8x8 (ref othello.cpp:14-262)
// ---------------------------------------------------------------------------------------------
class Othello final : public GameEnv {
public:
    explicit Othello(int n) : n_(n)
    {
        rot_ = rotationTables(n, n * n + 1);
        full_ = 0; not_left_ = 0; not_right_ = 0;
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                uint64_t b = 1ull << (y * n + x);
                full_ |= b;
                if (x != 0) { not_left_ |= b; }
                if (x != n - 1) { not_right_ |= b; }
            }
        reset();
    }
    std::unique_ptr<GameEnv> clone() const override { return std::make_unique<Othello>(*this); }
    void copyFrom(const GameEnv& o) override { *this = static_cast<const Othello&>(o); }
    void reset() override // ref othello.cpp:14-27 (black on init, init+n+1; white on init+1, init+n)
    {
        turn_ = 1;
        action_ids_.clear();
        action_players_.clear();
        const int init = n_ * (n_ / 2 - (1 - n_ % 2)) + (n_ / 2 - 1);
        s_[0] = (1ull << init) | (1ull << (init + n_ + 1));
        s_[1] = (1ull << (init + 1)) | (1ull << (init + n_));
    }
    // one step in direction d for every stone; stones that would leave the board vanish
    uint64_t shift(uint64_t b, int d) const
    {
        switch (d) {
            case 0: return (b << n_) & full_;                 // up
            case 1: return b >> n_;                           // down
            case 2: return (b & not_left_) >> 1;              // left
            case 3: return ((b & not_right_) << 1) & full_;   // right
            case 4: return ((b & not_left_) << (n_ - 1)) & full_;
            case 5: return ((b & not_right_) << (n_ + 1)) & full_;
            case 6: return (b & not_right_) >> (n_ - 1);
            default: return (b & not_left_) >> (n_ + 1);
        }
    }
    uint64_t moves(int player) const
    {
        const uint64_t me = s_[player - 1], op = s_[2 - player], empty = full_ & ~(me | op);
        uint64_t m = 0;
        for (int d = 0; d < 8; ++d) {
            uint64_t x = shift(me, d) & op;
            for (int i = 0; i < n_ - 3; ++i) { x |= shift(x, d) & op; }
            m |= shift(x, d) & empty;
        }
        return m;
    }
    bool isLegal(int a, int player) const override
    {
        const uint64_t m = moves(player);
        if (a == n_ * n_) { return m == 0; } // pass only when the mover has no move (ref othello.cpp:195-201)
        return a >= 0 && a < n_ * n_ && ((m >> a) & 1);
    }
    bool act(int a, int player) override
    {
        if (!isLegal(a, player)) { return false; }
        actUnchecked(a, player);
        return true;
    }
    void actUnchecked(int a, int player) override
    {
        action_ids_.push_back(static_cast<int16_t>(a));
        action_players_.push_back(static_cast<uint8_t>(player));
        turn_ = 3 - player;
        if (a == n_ * n_) { return; }
        uint64_t& me = s_[player - 1];
        uint64_t& op = s_[2 - player];
        const uint64_t placed = 1ull << a;
        uint64_t flip = 0;
        for (int d = 0; d < 8; ++d) {
            uint64_t line = 0, x = shift(placed, d);
            while (x & op) { line |= x; x = shift(x, d); }
            if (x & me) { flip |= line; }
        }
        me |= placed | flip;
        op &= ~flip;
    }
    void legalMask(uint8_t* out) const override
    {
        const uint64_t m = moves(turn_);
        for (int a = 0; a < n_ * n_; ++a) { out[a] = (m >> a) & 1; }
        out[n_ * n_] = (m == 0);
    }
    bool isTerminal() const override
    {
        const size_t k = action_ids_.size();
        return k >= 2 && action_ids_[k - 1] == n_ * n_ && action_ids_[k - 2] == n_ * n_;
    }
    float evalScore(bool is_resign) const override // ref othello.cpp:211-236: winner only once neither side can move
    {
        if (is_resign) { return scoreOf(3 - turn_); }
        if (moves(1) != 0 || moves(2) != 0) { return 0.0f; }
        const int b = __builtin_popcountll(s_[0]), w = __builtin_popcountll(s_[1]);
        return scoreOf(b > w ? 1 : (b < w ? 2 : 0));
    }
    void features(int r, float* out) const override
    {
        const int P = n_ * n_;
        const uint64_t own = s_[turn_ - 1], opp = s_[2 - turn_];
        const int* map = rot_->inv[r].data();
        for (int p = 0; p < P; ++p) {
            out[p] = (own >> map[p]) & 1 ? 1.0f : 0.0f;
            out[P + p] = (opp >> map[p]) & 1 ? 1.0f : 0.0f;
            out[2 * P + p] = turn_ == 1 ? 1.0f : 0.0f;
            out[3 * P + p] = turn_ == 2 ? 1.0f : 0.0f;
        }
    }
    bool hasDeviceTwin() const override { return n_ <= 8; }
    int deviceKind() const override { return 1; }
    void exportDeviceRoot(void* dst) const override // the fields of GoRootSnapshot the Othello device engine reads (go_body.h othLeafBody)
    {
        GoRootSnapshot& s = *static_cast<GoRootSnapshot*>(dst);
        s.stones[0][0] = s_[0];
        s.stones[1][0] = s_[1];
        s.hash = 0;
        s.hist_len = 0;
        s.turn = turn_;
        s.nmoves = static_cast<int32_t>(action_ids_.size());
        int passes = 0;
        for (size_t k = action_ids_.size(); k > 0 && passes < 2 && action_ids_[k - 1] == n_ * n_; --k) { ++passes; }
        s.passes = passes;
    }
    int numInputChannels() const override { return 4; }
    int boardSize() const override { return n_; }
    int policySize() const override { return n_ * n_ + 1; }
    std::string name() const override { return "othello_" + std::to_string(n_) + "x" + std::to_string(n_); }
    std::vector<std::pair<std::string, std::string>> loaderTags() const override { return {{"SZ", std::to_string(n_)}}; }

private:
    int n_;
    uint64_t full_, not_left_, not_right_, s_[2];
};

// ---------------------------------------------------------------------------------------------
// Go (ref go.cpp:102-315,690-723)
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxN = 19, kMaxP = kMaxN * kMaxN, kWords = (kMaxP + 63) / 64;
struct Bits {
    uint64_t w[kWords];
    void clear() { memset(w, 0, sizeof(w)); }
    void set(int i) { w[i >> 6] |= 1ull << (i & 63); }
    void reset(int i) { w[i >> 6] &= ~(1ull << (i & 63)); }
    bool test(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
};
struct GoStatic {
    int n = 0;
    int16_t nbr[kMaxP][4];
    uint8_t nnbr[kMaxP];
    uint64_t key[2][kMaxP];
};
const GoStatic* goStatic(int n)
{
    static std::mutex mu;
    static std::map<int, std::unique_ptr<GoStatic>> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto& slot = cache[n];
    if (!slot) {
        slot = std::make_unique<GoStatic>();
        slot->n = n;
        // the superko test only needs "equal positions <=> equal keys": any good 64-bit Zobrist keys do
        // (the reference draws its keys from mt19937_64(0), go.cpp:19-43; the values themselves are unobservable)
        std::mt19937_64 gen(0x6d7a677075ULL);
        for (int p = 0; p < n * n; ++p) {
            slot->key[0][p] = gen();
            slot->key[1][p] = gen();
            int x = p % n, y = p / n, k = 0;
            if (y + 1 < n) { slot->nbr[p][k++] = static_cast<int16_t>(p + n); }
            if (x + 1 < n) { slot->nbr[p][k++] = static_cast<int16_t>(p + 1); }
            if (y - 1 >= 0) { slot->nbr[p][k++] = static_cast<int16_t>(p - n); }
            if (x - 1 >= 0) { slot->nbr[p][k++] = static_cast<int16_t>(p - 1); }
            slot->nnbr[p] = static_cast<uint8_t>(k);
        }
    }
    return slot.get();
}
} // namespace

class Go final : public GameEnv {
    static constexpr int kHashCap = 1024; // > 2 * (2*361 + 1) positions of the longest legal game at 19x19
public:
    // situational: env_go_ko_rule=situational — a position repeats only with the same player to move: the hash of every position also
    // carries a turn key on odd move counts (ref go.cpp:45-49,141,222); positional (the default): key 0
    Go(int n, float komi, bool situational = false) : n_(n), P_(n * n), komi_(komi), st_(goStatic(n)), turn_key_(situational ? 0x9e3779b97f4a7c15ULL : 0)
    {
        rot_ = rotationTables(n, n * n + 1);
        reset();
    }
    std::unique_ptr<GameEnv> clone() const override { return std::make_unique<Go>(*this); }
    void copyFrom(const GameEnv& o) override { *this = static_cast<const Go&>(o); }
    void reset() override
    {
        turn_ = 1;
        action_ids_.clear();
        action_players_.clear();
        memset(board_, 0, sizeof(board_));
        hash_ = 0;
        memset(seen_, 0, sizeof(seen_));
        seen_used_ = 0;
        hist_len_ = 0;
        for (auto& h : hist_) { h[0].clear(); h[1].clear(); }
        stones_[0].clear();
        stones_[1].clear();
    }

    // ---- superko set: open addressing, 0 = empty slot (a zero hash is stored as 1) ----
    static uint64_t norm(uint64_t h) { return h ? h : 1; }
    bool seen(uint64_t h) const
    {
        h = norm(h);
        for (uint32_t i = static_cast<uint32_t>(h) & (kHashCap - 1);; i = (i + 1) & (kHashCap - 1)) {
            if (seen_[i] == 0) { return false; }
            if (seen_[i] == h) { return true; }
        }
    }
    void remember(uint64_t h)
    {
        h = norm(h);
        for (uint32_t i = static_cast<uint32_t>(h) & (kHashCap - 1);; i = (i + 1) & (kHashCap - 1)) {
            if (seen_[i] == h) { return; }
            if (seen_[i] == 0) { seen_[i] = h; ++seen_used_; return; }
        }
    }

    // flood fill the group at `start`; returns liberty count (stops at `cap` liberties), stones in grp[], XOR of keys in *gh
    int group(int start, int cap, int16_t* grp, int* gsize, uint64_t* gh) const
    {
        uint8_t mark[kMaxP];
        memset(mark, 0, P_);
        const uint8_t color = board_[start];
        int head = 0, tail = 0, libs = 0;
        uint64_t h = 0;
        grp[tail++] = static_cast<int16_t>(start);
        mark[start] = 1;
        while (head < tail) {
            const int p = grp[head++];
            h ^= st_->key[color - 1][p];
            for (int k = 0; k < st_->nnbr[p]; ++k) {
                const int q = st_->nbr[p][k];
                if (mark[q]) { continue; }
                if (board_[q] == 0) {
                    mark[q] = 1;
                    if (++libs >= cap) { *gsize = tail; if (gh) { *gh = 0; } return libs; } // early exit: enough liberties known
                } else if (board_[q] == color) {
                    mark[q] = 1;
                    grp[tail++] = static_cast<int16_t>(q);
                }
            }
        }
        *gsize = tail;
        if (gh) { *gh = h; }
        return libs;
    }

    bool isLegal(int a, int player) const override // ref go.cpp:208-244
    {
        if (a == P_) { return true; }
        if (a < 0 || a > P_ || board_[a] != 0) { return false; }
        bool ok = false;
        uint64_t nh = hash_ ^ turn_key_ ^ st_->key[player - 1][a];
        int16_t grp[kMaxP];
        int16_t seen_rep[4];
        int nrep = 0;
        for (int k = 0; k < st_->nnbr[a]; ++k) {
            const int q = st_->nbr[a][k];
            if (board_[q] == 0) { ok = true; continue; }
            int gs = 0;
            uint64_t gh = 0;
            if (board_[q] == player) {
                if (!ok && group(q, 2, grp, &gs, nullptr) > 1) { ok = true; }
            } else {
                // an enemy block in atari is captured: each block once
                bool dup = false;
                const int libs = group(q, 2, grp, &gs, &gh);
                if (libs == 1) {
                    int rep = grp[0];
                    for (int i = 1; i < gs; ++i) { rep = grp[i] < rep ? grp[i] : rep; }
                    for (int i = 0; i < nrep; ++i) { dup |= (seen_rep[i] == rep); }
                    if (!dup) { seen_rep[nrep++] = static_cast<int16_t>(rep); nh ^= gh; }
                    ok = true;
                }
            }
        }
        return ok && !seen(nh);
    }

    void legalMask(uint8_t* out) const override
    {
        // one whole-board pass: group id, liberty count (saturated at 2) and key-XOR per group
        int16_t gid[kMaxP];
        uint8_t glibs[kMaxP];
        uint64_t ghash[kMaxP];
        for (int p = 0; p < P_; ++p) { gid[p] = -1; }
        int16_t grp[kMaxP];
        for (int p = 0; p < P_; ++p) {
            if (board_[p] == 0 || gid[p] >= 0) { continue; }
            int gs = 0;
            uint64_t gh = 0;
            // full fill (cap = infinity would also do; liberties beyond 2 are never needed, but the hash is)
            const int libs = group(p, kMaxP, grp, &gs, &gh);
            for (int i = 0; i < gs; ++i) { gid[grp[i]] = static_cast<int16_t>(p); }
            glibs[p] = static_cast<uint8_t>(libs > 2 ? 2 : libs);
            ghash[p] = gh;
        }
        const int player = turn_;
        for (int a = 0; a < P_; ++a) {
            if (board_[a] != 0) { out[a] = 0; continue; }
            bool ok = false;
            uint64_t nh = hash_ ^ turn_key_ ^ st_->key[player - 1][a];
            int16_t cap[4];
            int ncap = 0;
            for (int k = 0; k < st_->nnbr[a]; ++k) {
                const int q = st_->nbr[a][k];
                if (board_[q] == 0) { ok = true; continue; }
                const int g = gid[q];
                if (board_[q] == player) {
                    if (glibs[g] > 1) { ok = true; }
                } else if (glibs[g] == 1) {
                    bool dup = false;
                    for (int i = 0; i < ncap; ++i) { dup |= (cap[i] == g); }
                    if (!dup) { cap[ncap++] = static_cast<int16_t>(g); nh ^= ghash[g]; }
                    ok = true;
                }
            }
            out[a] = ok && !seen(nh);
        }
        out[P_] = 1;
    }

    bool act(int a, int player) override
    {
        if (!isLegal(a, player)) { return false; }
        actUnchecked(a, player);
        return true;
    }
    void actUnchecked(int a, int player) override // ref go.cpp:132-190 (observable effects only)
    {
        action_ids_.push_back(static_cast<int16_t>(a));
        action_players_.push_back(static_cast<uint8_t>(player));
        turn_ = 3 - player;
        hash_ ^= turn_key_;
        if (a != P_) {
            board_[a] = static_cast<uint8_t>(player);
            stones_[player - 1].set(a);
            hash_ ^= st_->key[player - 1][a];
            int16_t grp[kMaxP];
            for (int k = 0; k < st_->nnbr[a]; ++k) {
                const int q = st_->nbr[a][k];
                if (board_[q] != 3 - player) { continue; }
                int gs = 0;
                if (group(q, 1, grp, &gs, nullptr) == 0) { // captured
                    for (int i = 0; i < gs; ++i) {
                        const int s = grp[i];
                        board_[s] = 0;
                        stones_[2 - player].reset(s);
                        hash_ ^= st_->key[2 - player][s];
                    }
                }
            }
        }
        hist_[hist_len_ & 7][0] = stones_[0];
        hist_[hist_len_ & 7][1] = stones_[1];
        ++hist_len_;
        remember(hash_);
    }
    bool isTerminal() const override // ref go.cpp:246-257
    {
        const size_t k = action_ids_.size();
        if (k >= 2 && action_ids_[k - 1] == P_ && action_ids_[k - 2] == P_) { return true; }
        return static_cast<int>(k) > 2 * P_;
    }
    float evalScore(bool is_resign) const override // Tromp-Taylor area + komi (ref go.cpp:259-278,703-723)
    {
        if (is_resign) { return scoreOf(3 - turn_); }
        float t1 = 0, t2 = 0;
        for (int p = 0; p < P_; ++p) { t1 += board_[p] == 1; t2 += board_[p] == 2; }
        t2 += komi_;
        uint8_t mark[kMaxP];
        memset(mark, 0, P_);
        int16_t q[kMaxP];
        for (int s = 0; s < P_; ++s) {
            if (board_[s] != 0 || mark[s]) { continue; }
            int head = 0, tail = 0, border = 0; // border: bit0 = touches black, bit1 = touches white
            q[tail++] = static_cast<int16_t>(s);
            mark[s] = 1;
            while (head < tail) {
                const int p = q[head++];
                for (int k = 0; k < st_->nnbr[p]; ++k) {
                    const int r = st_->nbr[p][k];
                    if (board_[r] == 0) { if (!mark[r]) { mark[r] = 1; q[tail++] = static_cast<int16_t>(r); } }
                    else { border |= board_[r]; }
                }
            }
            // a region bordered by black only — or by nothing at all (empty board) — counts for black (go.cpp:714),
            // else one bordered by white only counts for white
            if ((border & 2) == 0) { t1 += tail; }
            else if ((border & 1) == 0) { t2 += tail; }
        }
        return scoreOf(t1 > t2 ? 1 : (t1 < t2 ? 2 : 0));
    }
    void features(int r, float* out) const override // ref go.cpp:280-308
    {
        const int* map = rot_->inv[r].data();
        for (int k = 0; k < 8; ++k) {
            float* own = out + (2 * k) * P_;
            float* opp = own + P_;
            if (hist_len_ - 1 - k < 0) {
                memset(own, 0, 2 * P_ * sizeof(float));
                continue;
            }
            const Bits* h = hist_[(hist_len_ - 1 - k) & 7];
            const Bits& mine = h[turn_ - 1];
            const Bits& theirs = h[2 - turn_];
            for (int p = 0; p < P_; ++p) {
                own[p] = mine.test(map[p]) ? 1.0f : 0.0f;
                opp[p] = theirs.test(map[p]) ? 1.0f : 0.0f;
            }
        }
        const float b = turn_ == 1 ? 1.0f : 0.0f, w = turn_ == 2 ? 1.0f : 0.0f;
        for (int p = 0; p < P_; ++p) { out[16 * P_ + p] = b; out[17 * P_ + p] = w; }
    }
    void featureBits(int r, uint32_t* out) const override
    {
        const int W32 = (P_ + 31) / 32;
        const int* map = rot_->inv[r].data();
        memset(out, 0, size_t(18) * W32 * sizeof(uint32_t));
        for (int k = 0; k < 8; ++k) {
            if (hist_len_ - 1 - k < 0) { break; }
            const Bits* h = hist_[(hist_len_ - 1 - k) & 7];
            const Bits& mine = h[turn_ - 1];
            const Bits& theirs = h[2 - turn_];
            uint32_t* own = out + (2 * k) * W32;
            uint32_t* opp = own + W32;
            for (int p = 0; p < P_; ++p) {
                const int q = map[p];
                own[p >> 5] |= static_cast<uint32_t>(mine.test(q)) << (p & 31);
                opp[p >> 5] |= static_cast<uint32_t>(theirs.test(q)) << (p & 31);
            }
        }
        uint32_t* t = out + (turn_ == 1 ? 16 : 17) * W32;
        for (int p = 0; p < P_; ++p) { t[p >> 5] |= 1u << (p & 31); }
    }
    bool hasDeviceTwin() const override { return true; }
    uint64_t turnKey() const override { return turn_key_; }
    const uint64_t* zobristKeys() const override
    {
        // [2][P_] contiguous copy of the [2][kMaxP] table
        static std::mutex mu;
        static std::map<int, std::vector<uint64_t>> cache;
        std::lock_guard<std::mutex> lock(mu);
        auto& k = cache[n_];
        if (k.empty()) {
            k.resize(size_t(2) * P_);
            for (int c = 0; c < 2; ++c) { for (int p = 0; p < P_; ++p) { k[size_t(c) * P_ + p] = st_->key[c][p]; } }
        }
        return k.data();
    }
    void exportDeviceRoot(void* dst) const override
    {
        static_assert(kWords == kGoMaxW && kHashCap == kGoSeenCap && kMaxP == kGoMaxP, "device snapshot layout");
        GoRootSnapshot& s = *static_cast<GoRootSnapshot*>(dst);
        memcpy(s.stones, stones_, sizeof(s.stones));
        memcpy(s.hist, hist_, sizeof(s.hist));
        memcpy(s.seen, seen_, sizeof(s.seen));
        s.hash = hash_;
        s.hist_len = hist_len_;
        s.turn = turn_;
        s.nmoves = static_cast<int32_t>(action_ids_.size());
        int passes = 0;
        for (size_t k = action_ids_.size(); k > 0 && passes < 2 && action_ids_[k - 1] == P_; --k) { ++passes; }
        s.passes = passes;
        // group id per stone: the first point of the group in scan order
        int16_t grp[kMaxP];
        uint8_t done[kMaxP];
        memset(done, 0, P_);
        for (int p = 0; p < P_; ++p) {
            if (board_[p] == 0 || done[p]) { s.lab[p] = board_[p] == 0 ? 0 : s.lab[p]; continue; }
            int gs = 0;
            (void)group(p, kMaxP + 1, grp, &gs, nullptr);
            for (int i = 0; i < gs; ++i) { s.lab[grp[i]] = static_cast<uint16_t>(p); done[grp[i]] = 1; }
        }
    }
    int numInputChannels() const override { return 18; }
    int boardSize() const override { return n_; }
    int policySize() const override { return P_ + 1; }
    std::string name() const override { return "go_" + std::to_string(n_) + "x" + std::to_string(n_); }
    std::vector<std::pair<std::string, std::string>> loaderTags() const override
    {
        return {{"SZ", std::to_string(n_)}, {"KM", std::to_string(komi_)}};
    }

private:
    int n_, P_;
    float komi_;
    const GoStatic* st_;
    uint64_t turn_key_;
    uint8_t board_[kMaxP];
    uint64_t hash_;
    uint64_t seen_[kHashCap];
    int seen_used_;
    Bits stones_[2];
    Bits hist_[8][2];
    int hist_len_;
};

// ---------------------------------------------------------------------------------------------
// Atari-shaped synthetic environment (BASELINE configs[4]; ALE, OpenCV and the ROMs are not available).
// Feature contract of the reference's AtariEnv (ref atari.h:17-27, atari.cpp:48-131): 1 player, 18 actions all legal,
// features = for the last 8 steps [1 plane action_id / 18, 3 planes RGB / 255 of a 96x96 screen], oldest first.
// Screens are kept as bytes in an 8-deep ring (27 KB each) and expanded to f32 only when the planes are written.
// ---------------------------------------------------------------------------------------------
class AtariSynth final : public GameEnv {
    static constexpr int kRes = 96, kHist = 8, kActions = 18, kFrame = 3 * kRes * kRes;
    static uint64_t mix(uint64_t z)
    {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
public:
    AtariSynth(const std::string& name, int episode_length, size_t recent_observations)
        : name_(name), episode_length_(episode_length), recent_(recent_observations), frames_(size_t(kHist) * kFrame, 0) { resetSeed(0); }
    std::unique_ptr<GameEnv> clone() const override { return std::make_unique<AtariSynth>(*this); }
    void copyFrom(const GameEnv& o) override { *this = static_cast<const AtariSynth&>(o); }
    bool needsSeed() const override { return true; }
    void reset() override { resetSeed(0); }
    void resetSeed(int seed) override
    {
        turn_ = 1;
        seed_ = seed;
        reward_ = total_reward_ = 0;
        action_ids_.clear();
        action_players_.clear();
        std::fill(frames_.begin(), frames_.end(), 0);
        for (auto& v : valid_) { v = false; }
        for (auto& a : action_plane_) { a = 0.0f; }
        head_ = 0;
        pending_.clear();
        pushed_ = 0;
        lives_.clear();
        lives_.push_back(livesAt(0)); // ref atari.cpp:61-62
        lost_ = 0;
        observations_.clear();
        pushFrame(0, 0.0f, false);
    }
    bool isLegal(int a, int) const override { return a >= 0 && a < kActions; }
    bool act(int a, int player) override
    {
        if (!isLegal(a, player)) { return false; }
        actUnchecked(a, player);
        return true;
    }
    void actUnchecked(int a, int player) override
    {
        const int step = static_cast<int>(action_ids_.size()) + 1;
        const uint64_t h = mix(static_cast<uint64_t>(static_cast<uint32_t>(seed_)) * 0x9E3779B97F4A7C15ULL + 0x5157ULL * step);
        reward_ = ((h >> 40) < uint64_t(0.05 * (1 << 24))) ? 1.0f : 0.0f;
        total_reward_ += reward_;
        lost_ += (mix(static_cast<uint64_t>(static_cast<uint32_t>(seed_)) * 0xA24BAED4963EE407ULL + 0x11F3ULL * step) % 23) == 0;
        lives_.push_back(lost_ >= 3 ? 0 : 3 - lost_); // ref atari.cpp:83: the synthetic ale_.lives()
        action_ids_.push_back(static_cast<int16_t>(a));
        action_players_.push_back(static_cast<uint8_t>(player));
        pushFrame(step, a * 1.0f / kActions, true);
    }
    void legalMask(uint8_t* out) const override { for (int a = 0; a < kActions; ++a) { out[a] = 1; } }
    bool isTerminal() const override { return static_cast<int>(action_ids_.size()) >= episode_length_; }
    float evalScore(bool) const override { return total_reward_; }
    float reward() const override { return reward_; }
    void features(int, float* out) const override
    {
        materialize();
        for (int i = 0; i < kHist; ++i) { // oldest first; slot = (head_ + i) % kHist
            const int slot = (head_ + i) % kHist;
            float* dst = out + size_t(i) * 4 * kRes * kRes;
            const float av = action_plane_[slot];
            for (int p = 0; p < kRes * kRes; ++p) { dst[p] = av; }
            float* rgb = dst + kRes * kRes;
            if (!valid_[slot]) {
                memset(rgb, 0, size_t(kFrame) * sizeof(float));
            } else {
                const uint8_t* f = frames_.data() + size_t(slot) * kFrame;
                for (int p = 0; p < kFrame; ++p) { rgb[p] = static_cast<float>(f[p]) / 255.0f; }
            }
        }
    }
    // raw form of features(): the 8 screens oldest first (3 x 96 x 96 bytes each), then 8 f32 action-plane values, then 8 valid bytes
    int rawFeatureBytes() const override { return kHist * kFrame + kHist * 4 + kHist; }
    void rawFeatures(uint8_t* dst) const override
    {
        materialize();
        float av[kHist];
        for (int i = 0; i < kHist; ++i) {
            const int slot = (head_ + i) % kHist;
            memcpy(dst + size_t(i) * kFrame, frames_.data() + size_t(slot) * kFrame, kFrame);
            av[i] = action_plane_[slot];
            dst[size_t(kHist) * kFrame + kHist * 4 + i] = valid_[slot] ? 1 : 0;
        }
        memcpy(dst + size_t(kHist) * kFrame, av, sizeof(av));
    }
    uint64_t rawSerial() const override { return serial_; }
    int rawValidCount() const override { return pushed_ < kHist ? pushed_ : kHist; } // (no screen is drawn for this)
    int rawFrameBytes() const override { return kFrame; }
    void rawNewest(uint8_t* frame, uint8_t* meta) const override
    {
        materialize();
        memcpy(frame, frames_.data() + size_t((head_ + kHist - 1) % kHist) * kFrame, kFrame);
        float av[kHist];
        for (int i = 0; i < kHist; ++i) {
            const int slot = (head_ + i) % kHist;
            av[i] = action_plane_[slot];
            meta[kHist * 4 + i] = valid_[slot] ? 1 : 0;
        }
        memcpy(meta, av, sizeof(av));
    }
    int numInputChannels() const override { return kHist * 4; }
    int boardSize() const override { return kRes; }
    int policySize() const override { return kActions; }
    int numPlayers() const override { return 1; }
    std::string name() const override { return "atari_" + name_; }
    int actionFromString(const std::string& str) const override // ref atari.cpp:9-39: ALE's action names without their PLAYER_A_ prefix, upper-cased
    {
        static const char* const kNames[18] = {"NOOP", "FIRE", "UP", "RIGHT", "LEFT", "DOWN", "UPRIGHT", "UPLEFT", "DOWNRIGHT", "DOWNLEFT", "UPFIRE", "RIGHTFIRE", "LEFTFIRE",
                                               "DOWNFIRE", "UPRIGHTFIRE", "UPLEFTFIRE", "DOWNRIGHTFIRE", "DOWNLEFTFIRE"};
        std::string up = str;
        for (char& c : up) { c = static_cast<char>(std::toupper(static_cast<unsigned char>(c))); }
        for (int a = 0; a < 18; ++a) { if (up == kNames[a]) { return a; } }
        return -1;
    }
    std::vector<std::pair<std::string, std::string>> loaderTags() const override { return {{"SD", std::to_string(seed_)}}; }
    bool hasObservations() const override { return true; }
    void appendObservations(std::string* out) const override { materialize(); for (const auto& o : observations_) { out->append(o); } }
    const std::vector<int>* livesHistory() const override { return &lives_; }

private:
    int livesAt(int) const { return 3; }
    // A step only notes which screen is due: the 27-KB screen and its observation string are drawn when somebody looks at them (the
    // worker's leaf / record builders, which run one game per thread), not inside the RNG-ordered serial section of a move, where the 64
    // games of a pool would draw theirs one after the other with the GPU waiting (0.3 ms of a 7-ms move on BASELINE configs[4]).
    struct PendingFrame { int step; float action_value; bool with_action; };
    void pushFrame(int step, float action_value, bool with_action)
    {
        pending_.push_back(PendingFrame{step, action_value, with_action});
        ++pushed_;
        ++serial_;
    }
    void materialize() const
    {
        if (pending_.empty()) { return; }
        AtariSynth* self = const_cast<AtariSynth*>(this); // the ring and the observation list are caches of (seed, steps)
        for (const PendingFrame& p : pending_) { self->drawFrame(p.step, p.action_value, p.with_action); }
        self->pending_.clear();
    }
    void drawFrame(int step, float action_value, bool with_action)
    {
        // ring of the last 8 (action, screen) pairs: overwrite the oldest slot, which then becomes the newest
        const int slot = head_;
        uint8_t* f = frames_.data() + size_t(slot) * kFrame;
        // the synthetic screen of (seed, step): one hash byte per 8x8 block of every colour plane — piecewise constant like a real frame
        const uint64_t base = static_cast<uint64_t>(static_cast<uint32_t>(seed_)) * 0xD1B54A32D192ED03ULL + static_cast<uint64_t>(step) * 0x100000001B3ULL;
        constexpr int kB = kRes / 8;
        for (int cy = 0; cy < 3 * kB; ++cy) { // (plane, block row)
            uint8_t row[kRes];
            for (int bx = 0; bx < kB; ++bx) {
                const uint8_t v = static_cast<uint8_t>(mix(base + static_cast<uint64_t>(cy * kB + bx) * 0x9E3779B97F4A7C15ULL) >> 56);
                memset(row + bx * 8, v, 8);
            }
            for (int y = 0; y < 8; ++y) { memcpy(f + (size_t(cy) * 8 + y) * kRes, row, kRes); }
        }
        // ref atari.cpp:85-91: the observation string of the step, older ones dropped beyond the configured window
        observations_.emplace_back(reinterpret_cast<const char*>(f), size_t(kFrame));
        if (observations_.size() > recent_) {
            std::string& old = observations_[observations_.size() - recent_];
            old.clear();
            old.shrink_to_fit();
        }
        valid_[slot] = true;
        action_plane_[slot] = with_action ? action_value : 0.0f;
        head_ = (head_ + 1) % kHist;
    }
    uint64_t serial_ = 0;
    mutable std::vector<PendingFrame> pending_;
    int pushed_ = 0;
    std::string name_;
    int episode_length_, seed_ = 0, head_ = 0, lost_ = 0;
    size_t recent_;
    float reward_ = 0, total_reward_ = 0;
    std::vector<int> lives_;
    std::vector<std::string> observations_;
    std::vector<uint8_t> frames_;
    bool valid_[kHist] = {};
    float action_plane_[kHist] = {};
};

std::unique_ptr<GameEnv> createGameEnv(const std::string& game, int board_size, float go_komi, const std::string& atari_name, int atari_episode_length,
                                       const std::string& go_ko_rule, size_t atari_recent_observations)
{
    if (game == "atari") { return std::make_unique<AtariSynth>(atari_name, atari_episode_length, std::max<size_t>(1, atari_recent_observations)); }
    if (game == "tictactoe") { return std::make_unique<TicTacToe>(); }
    if (game == "othello") {
        const int n = board_size > 0 ? board_size : 8;
        if (n < 4 || n > 8 || n % 2) { setError("othello board size %d not supported (even, 4..8)", n); return nullptr; }
        return std::make_unique<Othello>(n);
    }
    if (game == "go") {
        const int n = board_size > 0 ? board_size : 9;
        if (n < 2 || n > kMaxN) { setError("go board size %d not supported (2..19)", n); return nullptr; }
        if (go_ko_rule != "positional" && go_ko_rule != "situational") { setError("env_go_ko_rule '%s' not supported (positional | situational, ref go.cpp:47)", go_ko_rule.c_str()); return nullptr; }
        return std::make_unique<Go>(n, go_komi, go_ko_rule == "situational");
    }
    setError("unknown env_game '%s' (tictactoe | go | othello | atari)", game.c_str());
    return nullptr;
}

} // namespace mz
