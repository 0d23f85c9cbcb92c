// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Board-game rules + feature planes.
// PARITY UNPINNED (cannot compile the reference envs here: base_env.h:6 -> utils.h:4-6 needs Boost).
#include "oracle.h"
#include <algorithm>
#include <bitset>
#include <cassert>
#include <unordered_set>

namespace mzo {

// =====================================================================================
// TicTacToe — ref: environment/tictactoe/tictactoe.{h,cpp}
// =====================================================================================
class TicTacToeEnv : public Env {
public:
    TicTacToeEnv() { reset(); }
    std::unique_ptr<Env> clone() const override { return std::make_unique<TicTacToeEnv>(*this); }
    void reset() override // ref tictactoe.cpp:11-17
    {
        turn_ = kPlayer1;
        actions_.clear();
        board_.assign(9, kPlayerNone);
    }
    bool act(const Action& action) override // ref :19-26
    {
        if (!isLegalAction(action)) { return false; }
        actions_.push_back(action);
        board_[action.getActionID()] = action.getPlayer();
        turn_ = getNextPlayer(action.getPlayer(), 2);
        return true;
    }
    bool isLegalAction(const Action& action) const override // ref :44-49
    {
        return (action.getActionID() >= 0 && action.getActionID() < 9 && board_[action.getActionID()] == kPlayerNone);
    }
    bool isTerminal() const override // ref :51-55
    {
        return (eval() != kPlayerNone || std::find(board_.begin(), board_.end(), kPlayerNone) == board_.end());
    }
    float getEvalScore(bool is_resign = false) const override // ref :57-65
    {
        Player result = (is_resign ? getNextPlayer(turn_, 2) : eval());
        switch (result) {
            case kPlayer1: return 1.0f;
            case kPlayer2: return -1.0f;
            default: return 0.0f;
        }
    }
    std::vector<float> getFeatures(Rotation rotation) const override // ref :67-90
    {
        std::vector<float> features;
        for (int channel = 0; channel < 4; ++channel) {
            for (int pos = 0; pos < 9; ++pos) {
                int rotation_pos = getPositionByRotating(reversed_rotation[rotation], pos, 3);
                if (channel == 0) { features.push_back((board_[rotation_pos] == turn_ ? 1.0f : 0.0f)); }
                else if (channel == 1) { features.push_back((board_[rotation_pos] == getNextPlayer(turn_, 2) ? 1.0f : 0.0f)); }
                else if (channel == 2) { features.push_back((turn_ == kPlayer1 ? 1.0f : 0.0f)); }
                else if (channel == 3) { features.push_back((turn_ == kPlayer2 ? 1.0f : 0.0f)); }
            }
        }
        return features;
    }
    std::vector<float> getActionFeatures(const Action& action, Rotation rotation) const override // ref :92-97
    {
        std::vector<float> action_features(9, 0.0f);
        action_features[getRotateAction(action.getActionID(), rotation)] = 1.0f;
        return action_features;
    }
    int getNumInputChannels() const override { return 4; }
    int getBoardSize() const override { return 3; }
    int getPolicySize() const override { return 9; }
    std::string name() const override { return "tictactoe"; }
    std::vector<std::pair<std::string, std::string>> loaderTags() const override { return {{"SZ", "3"}}; }

private:
    Player eval() const // ref :120-146 (bitwise AND of the three Player ints)
    {
        int c;
        for (int i = 0; i < 3; ++i) {
            c = 3;
            for (int j = 0; j < 3; ++j) { c &= static_cast<int>(board_[i * 3 + j]); }
            if (c != kPlayerNone) { return static_cast<Player>(c); }
            c = 3;
            for (int j = 0; j < 3; ++j) { c &= static_cast<int>(board_[j * 3 + i]); }
            if (c != kPlayerNone) { return static_cast<Player>(c); }
        }
        c = 3;
        for (int i = 0; i < 3; ++i) { c &= static_cast<int>(board_[i * 3 + i]); }
        if (c != kPlayerNone) { return static_cast<Player>(c); }
        c = 3;
        for (int i = 0; i < 3; ++i) { c &= static_cast<int>(board_[i * 3 + (3 - 1 - i)]); }
        if (c != kPlayerNone) { return static_cast<Player>(c); }
        return kPlayerNone;
    }
    std::vector<Player> board_;
};

// =====================================================================================
// Othello — ref: environment/othello/othello.{h,cpp}
// =====================================================================================
class OthelloEnv : public Env {
    typedef std::bitset<256> BB; // ref othello.h:15-16 (kMaxOthelloBoardSize = 16)
public:
    explicit OthelloEnv(int board_size) : board_size_(board_size) { reset(); }
    std::unique_ptr<Env> clone() const override { return std::make_unique<OthelloEnv>(*this); }
    BB& get(BB (&pair)[2], Player p) { return pair[p == kPlayer1 ? 0 : 1]; }
    const BB& get(const BB (&pair)[2], Player p) const { return pair[p == kPlayer1 ? 0 : 1]; }

    void reset() override // ref othello.cpp:14-59
    {
        turn_ = kPlayer1;
        actions_.clear();
        legal_pass_[0] = legal_pass_[1] = false;
        board_[0].reset(); board_[1].reset();
        legal_board_[0].reset(); legal_board_[1].reset();
        one_board_.set();
        int init_place = board_size_ * (board_size_ / 2 - (1 - board_size_ % 2)) + (board_size_ / 2 - 1);
        Player opp = getNextPlayer(turn_, 2);
        get(board_, opp).set(init_place + 1, 1);
        get(board_, opp).set(init_place + board_size_, 1);
        get(board_, turn_).set(init_place, 1);
        get(board_, turn_).set(init_place + board_size_ + 1, 1);
        get(legal_board_, opp).set(init_place - 1, 1);
        get(legal_board_, opp).set(init_place - board_size_, 1);
        get(legal_board_, opp).set(init_place + board_size_ + 2, 1);
        get(legal_board_, opp).set(init_place + 2 * board_size_ + 1, 1);
        get(legal_board_, turn_).set(init_place + 2, 1);
        get(legal_board_, turn_).set(init_place - board_size_ + 1, 1);
        get(legal_board_, turn_).set(init_place + board_size_ - 1, 1);
        get(legal_board_, turn_).set(init_place + 2 * board_size_, 1);
        dir_step_[0] = board_size_;
        dir_step_[1] = -board_size_;
        dir_step_[2] = -1;
        dir_step_[3] = 1;
        dir_step_[4] = board_size_ - 1;
        dir_step_[5] = board_size_ + 1;
        dir_step_[6] = -board_size_ + 1;
        dir_step_[7] = -board_size_ - 1;
        for (auto& m : mask_) { m.reset(); }
        for (int i = 0; i < board_size_; i++) {
            for (int j = 0; j < board_size_; j++) {
                (i == 0 || i == board_size_ - 1) ? mask_[0].set(i * board_size_ + j, 0) : mask_[0].set(i * board_size_ + j, 1);
                (j == 0 || j == board_size_ - 1) ? mask_[2].set(i * board_size_ + j, 0) : mask_[2].set(i * board_size_ + j, 1);
                ((i == 0 || i == board_size_ - 1) || (j == 0 || j == board_size_ - 1)) ? mask_[4].set(i * board_size_ + j, 0) : mask_[4].set(i * board_size_ + j, 1);
            }
        }
        mask_[1] = mask_[0];
        mask_[3] = mask_[2];
        mask_[5] = mask_[4];
        mask_[6] = mask_[4];
        mask_[7] = mask_[4];
    }
    bool isPass(const Action& a) const { return a.getActionID() == board_size_ * board_size_; }
    bool act(const Action& action) override // ref :102-140
    {
        if (!isLegalAction(action)) { return false; }
        actions_.push_back(action);
        turn_ = getNextPlayer(action.getPlayer(), 2);
        if (isPass(action)) { return true; }
        Player player = action.getPlayer(), opp = getNextPlayer(player, 2);
        get(board_, player).set(action.getActionID(), 1);
        BB placed_pos, flip;
        placed_pos.set(action.getActionID(), 1);
        for (int i = 0; i < 8; i++) { flip |= getFlipPoint(dir_step_[i], mask_[i], placed_pos, get(board_, opp), get(board_, player)); }
        get(board_, player) |= flip;
        get(board_, opp) &= ~flip;
        BB empty_board = (one_board_ ^ (board_[0] | board_[1]));
        legal_board_[0].reset();
        legal_board_[1].reset();
        for (int i = 0; i < 8; i++) {
            get(legal_board_, player) |= getCanPutPoint(dir_step_[i], mask_[i], empty_board, get(board_, opp), get(board_, player));
            get(legal_board_, opp) |= getCanPutPoint(dir_step_[i], mask_[i], empty_board, get(board_, player), get(board_, opp));
        }
        legal_pass_[0] = legal_board_[0].none();
        legal_pass_[1] = legal_board_[1].none();
        return true;
    }
    bool isLegalAction(const Action& action) const override // ref :190-201
    {
        if (isPass(action)) { return legal_pass_[action.getPlayer() == kPlayer1 ? 0 : 1]; }
        return get(legal_board_, action.getPlayer())[action.getActionID()];
    }
    bool isTerminal() const override // ref :203-209
    {
        return (actions_.size() >= 2 && isPass(actions_.back()) && isPass(actions_[actions_.size() - 2]));
    }
    float getEvalScore(bool is_resign = false) const override // ref :211-236
    {
        Player result = (is_resign ? getNextPlayer(turn_, 2) : eval());
        switch (result) {
            case kPlayer1: return 1.0f;
            case kPlayer2: return -1.0f;
            default: return 0.0f;
        }
    }
    std::vector<float> getFeatures(Rotation rotation) const override // ref :237-255
    {
        std::vector<float> features;
        for (int channel = 0; channel < 4; ++channel) {
            for (int pos = 0; pos < board_size_ * board_size_; ++pos) {
                int rotation_pos = getPositionByRotating(reversed_rotation[rotation], pos, board_size_);
                if (channel == 0) { features.push_back((get(board_, turn_)[rotation_pos] == 1 ? 1.0f : 0.0f)); }
                else if (channel == 1) { features.push_back((get(board_, getNextPlayer(turn_, 2))[rotation_pos] == 1 ? 1.0f : 0.0f)); }
                else if (channel == 2) { features.push_back((turn_ == kPlayer1 ? 1.0f : 0.0f)); }
                else if (channel == 3) { features.push_back((turn_ == kPlayer2 ? 1.0f : 0.0f)); }
            }
        }
        return features;
    }
    std::vector<float> getActionFeatures(const Action& action, Rotation rotation) const override // ref :257-262
    {
        std::vector<float> action_features(board_size_ * board_size_, 0.0f);
        if (!isPass(action)) { action_features[getRotateAction(action.getActionID(), rotation)] = 1.0f; }
        return action_features;
    }
    int getNumInputChannels() const override { return 4; }
    int getBoardSize() const override { return board_size_; }
    int getPolicySize() const override { return board_size_ * board_size_ + 1; }
    std::string name() const override { return "othello_" + std::to_string(board_size_) + "x" + std::to_string(board_size_); }
    std::vector<std::pair<std::string, std::string>> loaderTags() const override { return {{"SZ", std::to_string(board_size_)}}; }

private:
    BB shift(int direction, const BB& c) const { return (direction > 0) ? (c << direction) : (c >> abs(direction)); } // ref :61-65
    BB getFlipPoint(int direction, BB mask, BB placed_pos, BB opponent_board, BB player_board) const // ref :67-85
    {
        BB candidate, tmp_flip, moves;
        candidate = opponent_board & shift(direction, placed_pos) & mask;
        while (candidate != 0) {
            tmp_flip |= candidate;
            candidate = shift(direction, candidate);
            moves = player_board & candidate;
            candidate = opponent_board & (candidate)&mask;
        }
        if (moves.none()) { tmp_flip.reset(); }
        return tmp_flip;
    }
    BB getCanPutPoint(int direction, BB mask, BB empty_board, BB opponent_board, BB player_board) const // ref :87-100
    {
        BB candidate, moves;
        candidate = opponent_board & shift(direction, player_board) & mask;
        while (candidate != 0) {
            moves |= empty_board & shift(direction, candidate);
            candidate = opponent_board & shift(direction, candidate) & mask;
        }
        return moves;
    }
    Player eval() const // ref :221-236
    {
        int total1 = board_[0].count(), total2 = board_[1].count();
        if (legal_board_[0].none() && legal_board_[1].none()) {
            if (total1 > total2) { return kPlayer1; }
            else if (total1 < total2) { return kPlayer2; }
            else { return kPlayerNone; }
        }
        return kPlayerNone;
    }
    int board_size_;
    int dir_step_[8];
    BB one_board_, mask_[8];
    bool legal_pass_[2];
    BB legal_board_[2], board_[2];
};

// =====================================================================================
// Go — ref: environment/go/go.{h,cpp}, go_block.h, go_grid.h.
// areas_/benson_ maintenance (go.cpp:184-186,464-676) is unobservable through legality,
// terminal, score, features or records and is omitted.
// =====================================================================================
typedef std::bitset<361> GoBB;
typedef uint64_t GoHashKey;
static std::vector<GoHashKey> g_grid_key[2];
static GoHashKey g_turn_key = 0;
static void goInitialize() // ref go.cpp:19-43: mt19937_64(0): turn key, then per position (empty, black, white)
{
    if (!g_grid_key[0].empty()) { return; }
    std::mt19937_64 generator;
    generator.seed(0);
    g_turn_key = generator(); // turn_hash_key: only enters the hashes with env_go_ko_rule=situational (go.cpp:45-49)
    g_grid_key[0].resize(361);
    g_grid_key[1].resize(361);
    for (int pos = 0; pos < 361; ++pos) {
        (void)generator(); // empty_hash_key[pos]
        g_grid_key[0][pos] = generator();
        g_grid_key[1][pos] = generator();
    }
}

class GoEnv : public Env {
    struct Block { // ref go_block.h
        int num_grid = 0, num_liberty = 0;
        Player player = kPlayerNone;
        GoHashKey hash_key = 0;
        GoBB grid_bb, liberty_bb;
        void reset() { *this = Block(); }
        void addLiberty(int pos) { if (liberty_bb.test(pos)) { return; } liberty_bb.set(pos); ++num_liberty; }
        void removeLiberty(int pos) { if (!liberty_bb.test(pos)) { return; } liberty_bb.reset(pos); --num_liberty; }
    };
public:
    GoEnv(int board_size, float komi, bool situational = false) : board_size_(board_size), cfg_komi_(komi)
    {
        goInitialize();
        turn_key_ = situational ? g_turn_key : 0; // getGoTurnHashKey() (go.cpp:45-49)
        int n = board_size_ * board_size_;
        grid_player_.resize(n); grid_block_.resize(n); blocks_.resize(n); neighbors_.resize(n);
        for (int pos = 0; pos < n; ++pos) { // ref go_grid.h:43-54: up(+y), right(+x), down(-y), left(-x)
            const int directions[4] = {0, 1, 0, -1};
            int x = pos % board_size_, y = pos / board_size_;
            for (int i = 0; i < 4; ++i) {
                int nx = x + directions[i], ny = y + directions[(i + 1) % 4];
                if (nx < 0 || nx >= board_size_ || ny < 0 || ny >= board_size_) { continue; }
                neighbors_[pos].push_back(ny * board_size_ + nx);
            }
        }
        reset();
    }
    std::unique_ptr<Env> clone() const override { return std::make_unique<GoEnv>(*this); }
    void reset() override // ref go.cpp:102-130
    {
        komi_ = cfg_komi_;
        turn_ = kPlayer1;
        hash_key_ = 0;
        stone_bb_[0].reset(); stone_bb_[1].reset();
        board_mask_.reset(); left_boundary_.reset(); right_boundary_.reset(); free_block_id_.reset();
        for (int i = 0; i < board_size_ * board_size_; ++i) {
            grid_player_[i] = kPlayerNone;
            grid_block_[i] = -1;
            blocks_[i].reset();
            board_mask_.set(i);
        }
        free_block_id_ = ~free_block_id_ & board_mask_;
        for (int i = 0; i < board_size_; ++i) {
            left_boundary_.set(i * board_size_);
            right_boundary_.set(i * board_size_ + (board_size_ - 1));
        }
        actions_.clear();
        stone_history_.clear();
        hash_table_.clear();
    }
    bool isPass(const Action& a) const { return a.getActionID() == board_size_ * board_size_; }
    bool act(const Action& action) override // ref go.cpp:132-190
    {
        if (!isLegalAction(action)) { return false; }
        const int position = action.getActionID();
        const Player player = action.getPlayer();
        turn_ = getNextPlayer(player, 2);
        hash_key_ ^= turn_key_; // go.cpp:141
        actions_.push_back(action);
        if (isPass(action)) {
            stone_history_.push_back({stone_bb_[0], stone_bb_[1]});
            hash_table_.insert(hash_key_);
            return true;
        }
        grid_player_[position] = player;
        hash_key_ ^= key(position, player);
        int nb = newBlock();
        grid_block_[position] = nb;
        blocks_[nb].player = player;
        blocks_[nb].num_grid = 1;
        blocks_[nb].grid_bb.set(position);
        blocks_[nb].hash_key ^= key(position, player);
        for (int neighbor_pos : neighbors_[position]) {
            if (grid_player_[neighbor_pos] == kPlayerNone) {
                blocks_[nb].addLiberty(neighbor_pos);
            } else {
                int ob = grid_block_[neighbor_pos];
                blocks_[ob].removeLiberty(position);
                if (blocks_[ob].player == player) {
                    nb = combineBlocks(nb, ob);
                } else if (blocks_[ob].num_liberty == 0) {
                    removeBlockFromBoard(ob);
                }
            }
        }
        stone_bb_[player == kPlayer1 ? 0 : 1] |= blocks_[nb].grid_bb;
        stone_history_.push_back({stone_bb_[0], stone_bb_[1]});
        hash_table_.insert(hash_key_);
        return true;
    }
    bool isLegalAction(const Action& action) const override // ref go.cpp:208-244
    {
        if (isPass(action)) { return true; }
        const int position = action.getActionID();
        const Player player = action.getPlayer();
        if (grid_player_[position] != kPlayerNone) { return false; }
        bool is_legal = false;
        GoBB checked;
        GoHashKey new_hash_key = hash_key_ ^ turn_key_ ^ key(position, player); // go.cpp:222
        for (int neighbor_pos : neighbors_[position]) {
            if (grid_player_[neighbor_pos] == kPlayerNone) {
                is_legal = true;
            } else {
                int b = grid_block_[neighbor_pos];
                if (checked.test(b)) { continue; }
                checked.set(b);
                if (blocks_[b].player == player) {
                    if (blocks_[b].num_liberty > 1) { is_legal = true; }
                } else if (blocks_[b].num_liberty == 1) {
                    new_hash_key ^= blocks_[b].hash_key;
                    is_legal = true;
                }
            }
        }
        return (is_legal && hash_table_.count(new_hash_key) == 0);
    }
    bool isTerminal() const override // ref go.cpp:246-257
    {
        if (actions_.size() >= 2 && isPass(actions_.back()) && isPass(actions_[actions_.size() - 2])) { return true; }
        if (static_cast<int>(actions_.size()) > 2 * board_size_ * board_size_) { return true; }
        return false;
    }
    float getEvalScore(bool is_resign = false) const override // ref go.cpp:259-278
    {
        Player eval;
        if (is_resign) {
            eval = getNextPlayer(turn_, 2);
        } else {
            float t1, t2;
            calculateTrompTaylorTerritory(t1, t2);
            eval = (t1 > t2) ? kPlayer1 : ((t1 < t2) ? kPlayer2 : kPlayerNone);
        }
        switch (eval) {
            case kPlayer1: return 1.0f;
            case kPlayer2: return -1.0f;
            default: return 0.0f;
        }
    }
    std::vector<float> getFeatures(Rotation rotation) const override // ref go.cpp:280-308
    {
        std::vector<float> features;
        for (int channel = 0; channel < 18; ++channel) {
            for (int pos = 0; pos < board_size_ * board_size_; ++pos) {
                int rotation_pos = getPositionByRotating(reversed_rotation[rotation], pos, board_size_);
                if (channel < 16) {
                    int last_n_turn = static_cast<int>(stone_history_.size()) - 1 - channel / 2;
                    if (last_n_turn < 0) {
                        features.push_back(0.0f);
                    } else {
                        Player player = (channel % 2 == 0 ? turn_ : getNextPlayer(turn_, 2));
                        const GoBB& bb = (player == kPlayer1 ? stone_history_[last_n_turn].first : stone_history_[last_n_turn].second);
                        features.push_back(bb.test(rotation_pos) ? 1.0f : 0.0f);
                    }
                } else if (channel == 16) {
                    features.push_back((turn_ == kPlayer1 ? 1.0f : 0.0f));
                } else if (channel == 17) {
                    features.push_back((turn_ == kPlayer2 ? 1.0f : 0.0f));
                }
            }
        }
        return features;
    }
    std::vector<float> getActionFeatures(const Action& action, Rotation rotation) const override // ref go.cpp:310-315
    {
        std::vector<float> action_features(board_size_ * board_size_, 0.0f);
        if (!isPass(action)) { action_features[getRotateAction(action.getActionID(), rotation)] = 1.0f; }
        return action_features;
    }
    int getNumInputChannels() const override { return 18; }
    int getBoardSize() const override { return board_size_; }
    int getPolicySize() const override { return board_size_ * board_size_ + 1; }
    std::string name() const override { return "go_" + std::to_string(board_size_) + "x" + std::to_string(board_size_); }
    std::vector<std::pair<std::string, std::string>> loaderTags() const override // ref base_env.h:363-367, go.h:129-133
    {
        return {{"SZ", std::to_string(board_size_)}, {"KM", std::to_string(komi_)}};
    }

private:
    static GoHashKey key(int pos, Player p) { return g_grid_key[p == kPlayer1 ? 0 : 1][pos]; }
    int newBlock() // ref go.cpp:373-379 (_Find_first)
    {
        int id = free_block_id_._Find_first();
        free_block_id_.reset(id);
        return id;
    }
    void removeBlock(int b) { free_block_id_.set(b); blocks_[b].reset(); }
    void removeBlockFromBoard(int b) // ref go.cpp:388-433 (area bookkeeping omitted)
    {
        Block& block = blocks_[b];
        GoBB grid_bitboard = block.grid_bb;
        while (!grid_bitboard.none()) {
            int pos = grid_bitboard._Find_first();
            grid_bitboard.reset(pos);
            grid_player_[pos] = kPlayerNone;
            grid_block_[pos] = -1;
            for (int neighbor_pos : neighbors_[pos]) {
                if (grid_player_[neighbor_pos] != getNextPlayer(block.player, 2)) { continue; }
                blocks_[grid_block_[neighbor_pos]].addLiberty(pos);
            }
        }
        hash_key_ ^= block.hash_key;
        stone_bb_[block.player == kPlayer1 ? 0 : 1] &= ~block.grid_bb;
        removeBlock(b);
    }
    int combineBlocks(int b1, int b2) // ref go.cpp:435-462, go_block.h:25-34
    {
        if (b1 == b2) { return b1; }
        if (blocks_[b1].num_grid < blocks_[b2].num_grid) { return combineBlocks(b2, b1); }
        GoBB grid_bitboard = blocks_[b2].grid_bb;
        while (!grid_bitboard.none()) {
            int pos = grid_bitboard._Find_first();
            grid_bitboard.reset(pos);
            grid_block_[pos] = b1;
        }
        blocks_[b1].hash_key ^= blocks_[b2].hash_key;
        blocks_[b1].grid_bb |= blocks_[b2].grid_bb;
        blocks_[b1].num_grid += blocks_[b2].num_grid;
        blocks_[b1].liberty_bb |= blocks_[b2].liberty_bb;
        blocks_[b1].num_liberty = blocks_[b1].liberty_bb.count();
        removeBlock(b2);
        return b1;
    }
    GoBB dilate(const GoBB& bb) const // ref go.cpp:351-359
    {
        return ((bb << board_size_) | (bb >> board_size_) | ((bb & ~left_boundary_) >> 1) | ((bb & ~right_boundary_) << 1) | bb) & board_mask_;
    }
    GoBB floodFill(int start, const GoBB& boundary) const // ref go.cpp:690-701
    {
        GoBB ff;
        ff.set(start);
        bool need_dilate = true;
        while (need_dilate) {
            GoBB d = dilate(ff) & boundary;
            need_dilate = (ff != d);
            ff = d;
        }
        return ff;
    }
    void calculateTrompTaylorTerritory(float& t1, float& t2) const // ref go.cpp:703-723
    {
        t1 = stone_bb_[0].count();
        t2 = stone_bb_[1].count() + komi_;
        GoBB empty = ~(stone_bb_[0] | stone_bb_[1]) & board_mask_;
        while (!empty.none()) {
            int pos = empty._Find_first();
            GoBB ff = floodFill(pos, empty);
            GoBB surrounding = dilate(ff) & ~ff;
            if ((surrounding & ~stone_bb_[0]).none()) { t1 += ff.count(); }
            else if ((surrounding & ~stone_bb_[1]).none()) { t2 += ff.count(); }
            empty &= ~ff;
        }
    }
    int board_size_;
    float cfg_komi_, komi_ = 7.5f;
    GoHashKey hash_key_ = 0, turn_key_ = 0;
    GoBB board_mask_, left_boundary_, right_boundary_, free_block_id_;
    GoBB stone_bb_[2];
    std::vector<Player> grid_player_;
    std::vector<int> grid_block_;
    std::vector<Block> blocks_;
    std::vector<std::vector<int>> neighbors_;
    std::vector<std::pair<GoBB, GoBB>> stone_history_;
    std::unordered_set<GoHashKey> hash_table_;
};

// =====================================================================================
// Atari-shaped synthetic environment (ALE / OpenCV / ROMs are absent: SURVEY.md §8d C5).  Keeps the reference's
// feature contract (ref atari.h:17-27, atari.cpp:48-131): 1 player, 18 actions all legal, features = for the last 8
// steps [1 plane action_id/18, 3 planes RGB/255 of a 96x96 screen], oldest first; getActionFeatures = 18x6x6 with
// plane action_id all ones; eval score = cumulative reward.  Screen bytes and rewards are a counter hash of
// (seed, step); episodes last env_atari_episode_length steps.
// =====================================================================================
static inline uint64_t amix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
class AtariSynthEnv : public Env {
public:
    static constexpr int kRes = 96, kHist = 8, kActions = 18;
    AtariSynthEnv(const Config& cfg, Random* rng) : name_(cfg.env_atari_name), episode_length_(cfg.env_atari_episode_length)
    {
        // ref atari.cpp:87: observations older than this are dropped (kAtariMaxNumFramesPerEpisode = 108000 when no intermediate sequences)
        recent_observation_length_ = static_cast<size_t>(cfg.zero_actor_intermediate_sequence_length == 0
                                                              ? 108000
                                                              : cfg.zero_actor_intermediate_sequence_length + kHist + cfg.learner_n_step_return + cfg.learner_muzero_unrolling_step) + 1;
        rng_ = rng;
        reset(); // ref atari.h:47-50: the constructor resets (and draws a seed)
    }
    std::unique_ptr<Env> clone() const override { return std::make_unique<AtariSynthEnv>(*this); }
    void reset() override { resetSeed(rng_ ? rng_->randInt() : 0); } // ref atari.h:54
    void resetWithSeed(int seed) override { resetSeed(seed); }
    void resetSeed(int seed) // ref atari.cpp:48-72
    {
        turn_ = kPlayer1;
        seed_ = seed;
        reward_ = 0;
        total_reward_ = 0;
        actions_.clear();
        lives_history_.clear();
        lives_history_.push_back(livesAt(0));            // ref atari.cpp:61-62
        observations_.clear();
        observations_.push_back(observationString(0));   // ref atari.cpp:64-66: the initial observation
        feature_history_.assign(kHist, std::vector<float>(3 * kRes * kRes, 0.0f));
        feature_history_.push_back(observation(0));
        feature_history_.erase(feature_history_.begin());
        action_feature_history_.assign(kHist, 0.0f);
    }
    bool act(const Action& action) override // ref atari.cpp:74-99
    {
        const int step = static_cast<int>(actions_.size()) + 1;
        reward_ = ((amix(static_cast<uint64_t>(static_cast<uint32_t>(seed_)) * 0x9E3779B97F4A7C15ULL + 0x5157ULL * step) >> 40) < uint64_t(0.05 * (1 << 24))) ? 1.0f : 0.0f;
        total_reward_ += reward_;
        lives_history_.push_back(livesAt(step));          // ref atari.cpp:83
        actions_.push_back(action);
        observations_.push_back(observationString(step)); // ref atari.cpp:85-91
        if (observations_.size() > recent_observation_length_) {
            observations_[observations_.size() - recent_observation_length_].clear();
            observations_[observations_.size() - recent_observation_length_].shrink_to_fit();
        }
        action_feature_history_.push_back(action.getActionID() * 1.0f / kActions);
        action_feature_history_.erase(action_feature_history_.begin());
        feature_history_.push_back(observation(step));
        feature_history_.erase(feature_history_.begin());
        return true;
    }
    bool isLegalAction(const Action& a) const override { return a.getActionID() >= 0 && a.getActionID() < kActions; }
    bool isTerminal() const override { return static_cast<int>(actions_.size()) >= episode_length_; }
    float getReward() const override { return reward_; }
    float getEvalScore(bool = false) const override { return total_reward_; }
    std::vector<float> getFeatures(Rotation) const override // ref atari.cpp:112-122
    {
        std::vector<float> f;
        f.reserve(size_t(kHist) * 4 * kRes * kRes);
        for (int i = 0; i < kHist; ++i) {
            f.insert(f.end(), size_t(kRes) * kRes, action_feature_history_[i]);
            f.insert(f.end(), feature_history_[i].begin(), feature_history_[i].end());
        }
        return f;
    }
    std::vector<float> getActionFeatures(const Action& action, Rotation) const override // ref atari.cpp:124-130
    {
        std::vector<float> f(size_t(kActions) * 36, 0.0f);
        std::fill(f.begin() + action.getActionID() * 36, f.begin() + (action.getActionID() + 1) * 36, 1.0f);
        return f;
    }
    int getNumInputChannels() const override { return kHist * 4; }
    int getBoardSize() const override { return kRes; }
    int getPolicySize() const override { return kActions; }
    int getNumPlayer() const override { return 1; }
    std::string name() const override { return "atari_" + name_; }
    std::vector<std::pair<std::string, std::string>> loaderTags() const override { return {{"SD", std::to_string(seed_)}}; } // ref atari.cpp:190
    const std::vector<std::string>& getObservationHistory() const override { return observations_; }
    std::vector<int> getLivesHistory() const override { return lives_history_; }

private:
    // The synthetic screen of (seed, step): every 8x8 block of every colour plane is one byte of a counter hash — piecewise-constant like
    // a real Atari frame (which gzip shrinks ~50x), not noise.
    uint8_t screenByte(int step, int c, int y, int x) const
    {
        const uint64_t base = static_cast<uint64_t>(static_cast<uint32_t>(seed_)) * 0xD1B54A32D192ED03ULL + static_cast<uint64_t>(step) * 0x100000001B3ULL;
        const uint64_t block = static_cast<uint64_t>((c * (kRes / 8) + y / 8) * (kRes / 8) + x / 8);
        return static_cast<uint8_t>(amix(base + block * 0x9E3779B97F4A7C15ULL) >> 56);
    }
    std::vector<float> observation(int step) const // bytes / 255 (ref atari.cpp:142-158 getObservation(scale_01))
    {
        std::vector<float> o(size_t(3) * kRes * kRes);
        for (int c = 0; c < 3; ++c) { for (int y = 0; y < kRes; ++y) { for (int x = 0; x < kRes; ++x) { o[(size_t(c) * kRes + y) * kRes + x] = static_cast<float>(screenByte(step, c, y, x)) / 255.0f; } } }
        return o;
    }
    std::string observationString(int step) const // ref atari.cpp:160-169: the same screen as chw bytes
    {
        std::string s(size_t(3) * kRes * kRes, '\0');
        for (int c = 0; c < 3; ++c) { for (int y = 0; y < kRes; ++y) { for (int x = 0; x < kRes; ++x) { s[(size_t(c) * kRes + y) * kRes + x] = static_cast<char>(screenByte(step, c, y, x)); } } }
        return s;
    }
    int livesAt(int step) const // synthetic ale_.lives(): 3 lives, one lost at every step whose counter hash hits 1 in 23
    {
        int lost = 0;
        for (int s = 1; s <= step; ++s) { lost += (amix(static_cast<uint64_t>(static_cast<uint32_t>(seed_)) * 0xA24BAED4963EE407ULL + 0x11F3ULL * s) % 23) == 0; }
        return lost >= 3 ? 0 : 3 - lost;
    }
    size_t recent_observation_length_ = 108001;
    std::vector<int> lives_history_;
    std::vector<std::string> observations_;
    std::string name_;
    int episode_length_, seed_ = 0;
    float reward_ = 0, total_reward_ = 0;
    std::vector<std::vector<float>> feature_history_;
    std::vector<float> action_feature_history_; // one scalar per step (the plane is constant)
};

std::unique_ptr<Env> createEnv(const Config& cfg, Random* rng)
{
    if (cfg.env_game == "atari") { return std::make_unique<AtariSynthEnv>(cfg, rng); }
    if (cfg.env_game == "tictactoe") { return std::make_unique<TicTacToeEnv>(); }
    if (cfg.env_game == "othello") { return std::make_unique<OthelloEnv>(cfg.env_board_size); }
    if (cfg.env_game == "go") { return std::make_unique<GoEnv>(cfg.env_board_size, cfg.env_go_komi, cfg.env_go_ko_rule == "situational"); }
    return nullptr;
}

} // namespace mzo
