// minizero::actor::{BaseActor, ZeroActor, createActor} facade over libmzgpu (ref actor/base_actor.h:16-55, actor/zero_actor.h:24-70,
// actor/create_actor.h:10-19): the per-actor surface the console / `think()` callers use, with the reference's method names, argument
// meaning and ownership.  One actor = one single-game worker in per-actor stepping mode (mz_manual_step=true, include/mzgpu.h): the search
// (selection, leaf evaluation on the MFMA tower, expansion, backup) runs inside the library.
//
// What is the reference's, member by member (every one of them is called by tests/csrc/facade_check.cpp on the GPU):
//   reset, resetSearch, act(Action), act(vector<string>), getRecord(tags), isEnvTerminal, getEvalScore, getEnvironment (const and non-const),
//   getNNEvaluationBatchIndex, getActionInfoHistory, think(with_play, display_board), beforeNNEvaluation, afterNNEvaluation, isSearchDone,
//   getSearchAction, isResign, getSearchInfo, setNetwork, createSearch; createActor(tree_node_size, network).
// What differs, and why (INTEGRATION.md §2 lists the same):
//   * afterNNEvaluation(output) does not consume `output`: the evaluation happens on the device inside the library, one
//     beforeNNEvaluation / afterNNEvaluation pair still = one simulation; getNNEvaluationBatchIndex() is 0 while one is in flight, -1 otherwise;
//   * think() runs the search as library calls of whole launches: actor_mcts_think_time_limit IS honoured (the clock is looked at between library calls of 1-64 cycles, a quarter of what still fits the limit, and the
//     decision is then taken from the simulations run so far, zero_actor.cpp:40-45), actor_mcts_think_batch_size and the virtual loss of ZeroActor::step
//     (ref zero_actor.cpp:128-157) are not — the self-play path (ActorGroup) never uses them;
//   * setNetwork(network) runs the actor ON the caller's network (mz_worker_create_shared): no second copy of the weights, a later
//     network->loadModel(...) is what the next search uses;
//   * getEnvironment() is an `Environment` over the library's rules engine with the members the actor's callers use (isTerminal, getTurn, setTurn,
//     getEvalScore, getReward, getFeatures, getRotateAction, isLegalAction, getLegalActions, getNumActions() for getActionHistory().size(), reset(seed), toString) —
//     toString() is a plain-text board, not the reference's ANSI-coloured one; game-specific members (GoEnv::getBenson..., ...) are not there;
//   * getSearchInfo() is a one-line summary (the reference's is a free-form, time-stamped debug string: zero_actor.cpp:159-176);
//   * createSearch() returns nullptr: the tree lives in the library's node pool (mz_pool_*).
// Header-only; link with -lmzgpu.
#pragma once
#include "mzgpu_config.h"
#include "network.h"
#include <chrono>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace minizero {

namespace utils {
enum class Rotation { // ref utils/rotation.h:9-19
    kRotationNone,
    kRotation90,
    kRotation180,
    kRotation270,
    kHorizontalRotation,
    kHorizontalRotation90,
    kHorizontalRotation180,
    kHorizontalRotation270,
    kRotateSize
};
} // namespace utils

namespace env {
enum class Player { kPlayerNone = 0, kPlayer1 = 1, kPlayer2 = 2, kPlayerSize = 3 }; // ref environment/base/base_env.h:13-18
inline char playerToChar(Player p) { return p == Player::kPlayerNone ? 'N' : p == Player::kPlayer1 ? 'B' : p == Player::kPlayer2 ? 'W' : '?'; } // ref base_env.cpp:5-13
inline Player charToPlayer(char c) // ref base_env.cpp:15-25
{
    switch (c) {
        case 'N': return Player::kPlayerNone;
        case 'B':
        case 'b': return Player::kPlayer1;
        case 'W':
        case 'w': return Player::kPlayer2;
        default: return Player::kPlayerSize;
    }
}
} // namespace env

// the two fields of the reference's game Actions the actor surface reads (ref environment/base/base_env.h:33-56)
class Action {
public:
    Action() : action_id_(-1), player_(env::Player::kPlayerNone) {}
    Action(int action_id, env::Player player) : action_id_(action_id), player_(player) {}
    inline int getActionID() const { return action_id_; }
    inline env::Player getPlayer() const { return player_; }

private:
    int action_id_;
    env::Player player_;
};

// What the actor's callers need from the game Environment (ref environment/base/base_env.h:74-114), over the rules engine inside the library
class Environment {
public:
    explicit Environment(mz_worker* const* w) : w_(w) {}
    inline bool isTerminal() const { return query(0) != 0.0f; }
    inline env::Player getTurn() const { return static_cast<env::Player>(static_cast<int>(query(1))); }
    inline void setTurn(env::Player p) { check(mz_worker_env_set_turn(worker(), 0, static_cast<int>(p))); }
    inline float getEvalScore(bool is_resign = false) const { return query(is_resign ? 3 : 2); }
    inline int getNumActions() const { return static_cast<int>(query(4)); } // == getActionHistory().size()
    inline float getReward() const { return query(5); }
    void reset(int seed) { check(mz_worker_env_reset_seed(worker(), 0, seed)); } // ref atari.h:55 (the console's load of a record with an SD tag)
    std::vector<float> getFeatures(utils::Rotation rotation = utils::Rotation::kRotationNone) const // ref base_env.h:88
    {
        const int n = mz_worker_env_features(worker(), 0, static_cast<int>(rotation), nullptr, 0);
        check(n);
        std::vector<float> f(static_cast<size_t>(n));
        check(mz_worker_env_features(worker(), 0, static_cast<int>(rotation), f.data(), n));
        return f;
    }
    int getRotateAction(int action_id, utils::Rotation rotation) const // ref base_env.h:99
    {
        const int r = mz_worker_env_rotate_action(worker(), 0, action_id, static_cast<int>(rotation));
        check(r);
        return r;
    }
    bool isLegalAction(const Action& action) const // ref base_env.h:84 (for the player to move)
    {
        const std::vector<uint8_t> m = legalMask();
        return action.getPlayer() == getTurn() && action.getActionID() >= 0 && action.getActionID() < static_cast<int>(m.size()) && m[action.getActionID()] != 0;
    }
    std::vector<Action> getLegalActions() const // ref base_env.h:83
    {
        const std::vector<uint8_t> m = legalMask();
        const env::Player p = getTurn();
        std::vector<Action> out;
        for (size_t a = 0; a < m.size(); ++a) { if (m[a]) { out.emplace_back(static_cast<int>(a), p); } }
        return out;
    }
    // plain-text board of the current position for the board games, from planes 0 / 1 of getFeatures() (own / opponent stones of the
    // player to move in every board game of the path: ref go.cpp:280-308, othello.cpp:237-255, tictactoe.cpp:67-90); X = black, O = white
    std::string toString() const // ref base_env.h:100
    {
        mz_net_desc d;
        check(mz_net_get_desc(mz_worker_net(worker()), &d));
        if (d.type == 2) { return "(observation screens)\n"; }
        const int board_size = d.input_channel_height, P = board_size * board_size;
        const std::vector<float> f = getFeatures();
        std::string s;
        if (static_cast<int>(f.size()) < 2 * P) { return s; }
        const bool black_to_move = getTurn() == env::Player::kPlayer1;
        for (int row = board_size - 1; row >= 0; --row) {
            for (int col = 0; col < board_size; ++col) {
                const int p = row * board_size + col;
                const bool own = f[p] != 0.0f, opp = f[P + p] != 0.0f;
                s += (own ? (black_to_move ? " X" : " O") : opp ? (black_to_move ? " O" : " X") : " .");
            }
            s += '\n';
        }
        return s;
    }

private:
    mz_worker* worker() const
    {
        if (!*w_) { std::cerr << "actor: setNetwork() has not been called" << std::endl; std::abort(); }
        return *w_;
    }
    static void check(int rc)
    {
        if (rc < 0) { std::cerr << mz_last_error() << std::endl; std::abort(); }
    }
    std::vector<uint8_t> legalMask() const
    {
        mz_net_desc d;
        check(mz_net_get_desc(mz_worker_net(worker()), &d));
        std::vector<uint8_t> m(static_cast<size_t>(d.action_size));
        check(mz_worker_env_legal_mask(worker(), 0, m.data(), static_cast<int>(m.size())));
        return m;
    }
    float query(int what) const
    {
        float v = 0.0f;
        check(mz_worker_env_query(worker(), 0, what, &v));
        return v;
    }
    mz_worker* const* w_;
};

namespace actor {

class Search { // ref actor/search.h
public:
    virtual ~Search() = default;
    virtual void reset() = 0;
};

class BaseActor {
public:
    BaseActor() : nn_evaluation_batch_id_(-1), env_(&worker_) {}
    virtual ~BaseActor() { if (worker_) { mz_worker_destroy(worker_); } }
    BaseActor(const BaseActor&) = delete;
    BaseActor& operator=(const BaseActor&) = delete;

    virtual void reset() // ref base_actor.cpp:8-13
    {
        check(mz_worker_reset_game(handle(), 0));
        resetSearch();
    }
    virtual void resetSearch() // ref base_actor.cpp:15-20
    {
        nn_evaluation_batch_id_ = -1;
        check(mz_worker_reset_search(handle()));
    }
    bool act(const Action& action) // ref base_actor.cpp:22-30
    {
        const int rc = mz_worker_act(handle(), 0, action.getActionID(), static_cast<int>(action.getPlayer()));
        check(rc);
        return rc == 1;
    }
    bool act(const std::vector<std::string>& action_string_args) // ref base_actor.cpp:32-40
    {
        std::vector<const char*> args;
        for (const auto& a : action_string_args) { args.push_back(a.c_str()); }
        const int rc = mz_worker_act_string(handle(), 0, args.data(), static_cast<int>(args.size()));
        check(rc);
        return rc == 1;
    }
    virtual std::string getRecord(const std::unordered_map<std::string, std::string>& tags = {}) const // ref base_actor.cpp:39-57
    {
        std::vector<const char*> k, v;
        for (const auto& t : tags) { k.push_back(t.first.c_str()); v.push_back(t.second.c_str()); }
        const int need = mz_worker_record(handle(), 0, k.data(), v.data(), static_cast<int>(k.size()), nullptr, 0); // Atari records: megabytes of OBS hex
        check(need);
        std::vector<char> buf(static_cast<size_t>(need) + 1);
        check(mz_worker_record(handle(), 0, k.data(), v.data(), static_cast<int>(k.size()), buf.data(), static_cast<int>(buf.size())));
        return buf.data();
    }

    inline bool isEnvTerminal() const { return env_.isTerminal(); }
    inline const float getEvalScore() const { return env_.getEvalScore(); }
    inline Environment& getEnvironment() { return env_; }
    inline const Environment& getEnvironment() const { return env_; }
    inline const int getNNEvaluationBatchIndex() const { return nn_evaluation_batch_id_; }
    // ref base_actor.h:33-34: per move its (key, value) pairs (P, V, R; L for the Atari-shaped game).  A snapshot of the library's history,
    // refreshed by every call.  The reference hands out its own member as a non-const reference (base_actor.h:33): code written against it
    // (`auto& h = actor->getActionInfoHistory(); h[i].clear();`) compiles here too, but writes into the snapshot do not reach the library's history —
    // the library clears the entries a record has emitted itself (actor_group.cpp:40-46)
    inline std::vector<std::vector<std::pair<std::string, std::string>>>& getActionInfoHistory() { return refreshActionInfoHistory(); }
    inline const std::vector<std::vector<std::pair<std::string, std::string>>>& getActionInfoHistory() const { return refreshActionInfoHistory(); }
    inline std::vector<std::vector<std::pair<std::string, std::string>>>& refreshActionInfoHistory() const
    {
        const int need = mz_worker_action_info_history(handle(), 0, nullptr, 0);
        check(need);
        std::vector<char> buf(static_cast<size_t>(need) + 1);
        check(mz_worker_action_info_history(handle(), 0, buf.data(), static_cast<int>(buf.size())));
        action_info_history_.assign(static_cast<size_t>(env_.getNumActions()), {});
        size_t move = 0;
        std::string field, key;
        bool have_key = false;
        for (int i = 0; i < need; ++i) {
            const char c = buf[i];
            if (c == '\x1e') { ++move; }
            else if (c == '\x1f') {
                if (!have_key) { key = field; have_key = true; }
                else { if (move < action_info_history_.size()) { action_info_history_[move].emplace_back(key, field); } have_key = false; }
                field.clear();
            } else { field += c; }
        }
        return action_info_history_;
    }

    virtual Action think(bool with_play = false, bool display_board = false) = 0;
    virtual void beforeNNEvaluation() = 0;
    virtual void afterNNEvaluation(const std::shared_ptr<network::NetworkOutput>& network_output) = 0;
    virtual bool isSearchDone() const = 0;
    virtual Action getSearchAction() const = 0;
    virtual bool isResign() const = 0;
    virtual std::string getSearchInfo() const = 0;
    virtual void setNetwork(const std::shared_ptr<network::Network>& network) = 0;
    virtual std::shared_ptr<Search> createSearch() = 0;

    inline mz_worker* handle() const
    {
        if (!worker_) { std::cerr << "actor: setNetwork() has not been called" << std::endl; std::abort(); }
        return worker_;
    }

protected:
    static void check(int rc)
    {
        if (rc < 0) { std::cerr << mz_last_error() << std::endl; std::abort(); }
    }
    int nn_evaluation_batch_id_;
    mz_worker* worker_ = nullptr;
    Environment env_;
    mutable std::vector<std::vector<std::pair<std::string, std::string>>> action_info_history_;
};

class ZeroActor : public BaseActor {
public:
    explicit ZeroActor(uint64_t tree_node_size) : tree_node_size_(tree_node_size) {}
    ~ZeroActor() override { if (worker_) { mz_worker_destroy(worker_); worker_ = nullptr; } } // before network_ lets go of the network the worker runs on

    // the resign coin of zero_actor.cpp:23-27 is drawn inside the library: by mz_worker_create for the first game (the reference draws it in
    // createActor's reset(), from the creating thread's generator, actor_group.cpp:179-187) and by mz_worker_reset_game afterwards
    void reset() override
    {
        if (fresh_) { fresh_ = false; resetSearch(); return; }
        BaseActor::reset();
    }
    void resetSearch() override
    {
        BaseActor::resetSearch();
        cycles_in_search_ = 0;
    }
    Action think(bool with_play = false, bool display_board = false) override // ref zero_actor.cpp:36-49
    {
        resetSearch();
        // actor_mcts_think_time_limit (seconds; 0 = none): the reference checks the clock after every step() = one batch of simulations; here a step is a chunk of
        // cycles (one launch), and when the limit breaks the loop the decision is taken from the simulations run so far (zero_actor.cpp:40-45)
        const std::string lim = config::mzgpuConfValue(config::mzgpuCollectConfiguration(), "actor_mcts_think_time_limit");
        const double limit_ms = lim.empty() ? 0.0 : std::atof(lim.c_str()) * 1000.0;
        const auto start = std::chrono::steady_clock::now();
        auto elapsed = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count(); };
        // Granularity of the limit: the clock is looked at between two library calls.  The first call runs kThinkFirst cycles (the root's expansion precedes any
        // decision), every later one a quarter of the cycles that still fit the limit at the rate measured so far (1 .. kThinkChunk): the search ends within
        // about a quarter of the remaining time of the limit, never a fixed 32 cycles late.
        int chunk = kThinkFirst, cycles_run = 0;
        while (!isSearchDone()) {
            if (limit_ms > 0) {
                const int ran = mz_worker_run_cycles(handle(), chunk);
                check(ran);
                cycles_run += ran;
                const double ms = elapsed();
                if (ms >= limit_ms) { break; }
                const double per_cycle = ms / (cycles_run > 0 ? cycles_run : 1);
                const double fit = per_cycle > 0 ? (limit_ms - ms) / per_cycle : kThinkChunk;
                chunk = fit / 4 < 1 ? 1 : (fit / 4 > kThinkChunk ? kThinkChunk : static_cast<int>(fit / 4));
            } else {
                step();
            }
        }
        if (!isSearchDone()) {
            // (a root that is not expanded yet — the limit fell inside the first two cycles — gets them: a decision needs the root's children)
            if (mz_worker_finish_search(handle()) == MZ_ERR_STATE) { check(mz_worker_run_cycles(handle(), 2)); if (!isSearchDone()) { check(mz_worker_finish_search(handle())); } }
        }
        const Action a = getSearchAction();
        if (with_play) { act(a); }
        if (display_board) { std::cerr << env_.toString() << getSearchInfo() << std::endl; }
        return a;
    }
    void beforeNNEvaluation() override { nn_evaluation_batch_id_ = 0; } // selection + leaf evaluation happen in afterNNEvaluation's cycle
    void afterNNEvaluation(const std::shared_ptr<network::NetworkOutput>& /*network_output: evaluated on the device, see the header comment*/) override
    {
        check(mz_worker_run_cycles(handle(), 1));
        nn_evaluation_batch_id_ = -1;
        // the (n + 1)-th evaluation completes the search: one more call consumes its output and takes the decision (zero_actor.cpp:96-97)
        if (++cycles_in_search_ == cyclesPerMove()) { check(mz_worker_run_cycles(handle(), 1)); }
    }
    bool isSearchDone() const override { return mz_worker_search_done(handle()) == 1; }
    Action getSearchAction() const override
    {
        int id = -1, player = 0, resign = 0;
        check(mz_worker_search_action(handle(), 0, &id, &player, &resign));
        return Action(id, static_cast<env::Player>(player));
    }
    bool isResign() const override
    {
        int id = -1, player = 0, resign = 0;
        check(mz_worker_search_action(handle(), 0, &id, &player, &resign));
        return resign != 0;
    }
    std::string getSearchInfo() const override // the fields of zero_actor.cpp:163-170 that do not need the tree (no time stamp, no node dumps)
    {
        const Action a = getSearchAction();
        std::ostringstream oss;
        oss << "model file name: " << (network_ ? network_->getNetworkFileName() : std::string()) << std::endl
            << "move number: " << env_.getNumActions() << ", action: " << a.getActionID() << ", reward: " << env_.getReward()
            << ", player: " << env::playerToChar(a.getPlayer()) << (isResign() ? " (resign)" : "") << std::endl;
        return oss.str();
    }
    void setNetwork(const std::shared_ptr<network::Network>& network) override // ref zero_actor.cpp:100-112
    {
        if (!network || !network->handle()) { std::cerr << "setNetwork: null network (call loadModel first)" << std::endl; std::abort(); }
        if (worker_ && network_ == network) { // ActorGroup's load_model path (actor_group.cpp:227-232): same object, new weights — follow the name
            check(mz_worker_command(worker_, ("load_model " + network->getNetworkFileName()).c_str()));
            return;
        }
        if (worker_) { mz_worker_destroy(worker_); worker_ = nullptr; }
        network_ = network; // shared ownership like the reference's alphazero_network_ / muzero_network_ members: the network outlives the worker
        std::string conf = config::mzgpuCollectConfiguration();
        // (n + 1) * action_size == tree_node_size (ref actor_group.cpp:183): the pool is sized from actor_num_simulation in the configuration
        conf += ":zero_num_parallel_games=1:mz_manual_step=true:nn_file_name=" + network->getNetworkFileName();
        worker_ = mz_worker_create_shared(network->getGPUID(), conf.c_str(), network->handle());
        if (!worker_) { std::cerr << mz_last_error() << std::endl; std::abort(); }
        check(mz_worker_command(worker_, "start"));
        fresh_ = true;
    }
    std::shared_ptr<Search> createSearch() override { return nullptr; } // the tree lives in the library's node pool (include/mzgpu.h mz_pool_*)
    inline uint64_t getTreeNodeSize() const { return tree_node_size_; }

protected:
    virtual void step() // ref zero_actor.cpp:128-150: here a whole search is one call (one kernel launch per move inside the library)
    {
        check(mz_worker_run_cycles(handle(), cyclesPerMove() + 1));
    }
    int cyclesPerMove() const { return mz_worker_cycles_per_move(handle()); }
    static constexpr int kThinkFirst = 2;  // cycles of the first call under a think-time limit (the root's expansion precedes any decision)
    static constexpr int kThinkChunk = 64; // ... and the most cycles between two looks at the clock
    uint64_t tree_node_size_;
    int cycles_in_search_ = 0;
    bool fresh_ = false; // the worker has just been created: its game is new
    std::shared_ptr<network::Network> network_;
};

// ref create_actor.h:10-19
inline std::shared_ptr<BaseActor> createActor(uint64_t tree_node_size, const std::shared_ptr<network::Network>& network)
{
    auto actor = std::make_shared<ZeroActor>(tree_node_size);
    actor->setNetwork(network);
    actor->reset();
    return actor;
}

} // namespace actor
} // namespace minizero
