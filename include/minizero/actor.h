// minizero::actor::{BaseActor, ZeroActor, createActor} facade over libmzgpu (ref actor/base_actor.h:16-55, actor/zero_actor.h:24-70,
// actor/create_actor.h:10-19): the per-actor surface the console / `think()` callers use, with the reference's method names, argument
// meaning and ownership.  One actor = one single-game worker in per-actor stepping mode (mz_manual_step=true, include/mzgpu.h): the search
// (selection, leaf evaluation on the MFMA tower, expansion, backup) runs inside the library, so
//   * beforeNNEvaluation() / afterNNEvaluation(output) keep their place in the caller's loop — one pair = one simulation — but the
//     network output argument is not consumed (the evaluation already happened on the device); getNNEvaluationBatchIndex() is 0 while a
//     simulation is in flight, -1 otherwise;
//   * think() = resetSearch() + simulations until isSearchDone() (+ act() when with_play), as zero_actor.cpp:36-49;
//   * setNetwork(network) takes the file name and GPU id of the minizero::network::Network facade; the worker reads the file itself.
// getEnvironment() returns a small read-only view (turn, terminal, eval score, action count): the reference's game-specific Environment
// classes stay on the caller's side of the boundary.  Header-only; link with -lmzgpu.
#pragma once
#include "mzgpu_config.h"
#include "network.h"
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace minizero {

namespace env {
enum class Player { kPlayerNone = 0, kPlayer1 = 1, kPlayer2 = 2 }; // ref environment/base/base_env.h:13-23
}

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

namespace actor {

class Search { // ref actor/search.h
public:
    virtual ~Search() = default;
    virtual void reset() = 0;
};

class EnvironmentView { // what the actor surface needs from Environment (ref base_env.h:74-114)
public:
    explicit EnvironmentView(mz_worker* const* w) : w_(w) {}
    inline bool isTerminal() const { return query(0) != 0.0f; }
    inline env::Player getTurn() const { return static_cast<env::Player>(static_cast<int>(query(1))); }
    inline float getEvalScore(bool is_resign = false) const { return query(is_resign ? 3 : 2); }
    inline int getNumActions() const { return static_cast<int>(query(4)); }
    inline float getReward() const { return query(5); }

private:
    float query(int what) const
    {
        float v = 0.0f;
        if (!*w_ || mz_worker_env_query(*w_, 0, what, &v) != MZ_OK) { std::cerr << mz_last_error() << std::endl; std::abort(); }
        return v;
    }
    mz_worker* const* w_;
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
    virtual std::string getRecord(const std::unordered_map<std::string, std::string>& tags = {}) const // ref base_actor.cpp:39-57
    {
        std::vector<const char*> k, v;
        for (const auto& t : tags) { k.push_back(t.first.c_str()); v.push_back(t.second.c_str()); }
        std::vector<char> buf(1 << 22);
        check(mz_worker_record(handle(), 0, k.data(), v.data(), static_cast<int>(k.size()), buf.data(), static_cast<int>(buf.size())));
        return buf.data();
    }

    inline bool isEnvTerminal() const { return env_.isTerminal(); }
    inline const float getEvalScore() const { return env_.getEvalScore(); }
    inline const EnvironmentView& getEnvironment() const { return env_; }
    inline const int getNNEvaluationBatchIndex() const { return nn_evaluation_batch_id_; }

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
    EnvironmentView env_;
};

class ZeroActor : public BaseActor {
public:
    explicit ZeroActor(uint64_t tree_node_size) : tree_node_size_(tree_node_size) {}

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
        while (!isSearchDone()) { step(); }
        const Action a = getSearchAction();
        if (with_play) { act(a); }
        if (display_board) { std::cerr << getSearchInfo() << std::endl; }
        return a;
    }
    void beforeNNEvaluation() override { nn_evaluation_batch_id_ = 0; } // selection + leaf evaluation happen in afterNNEvaluation's cycle
    void afterNNEvaluation(const std::shared_ptr<network::NetworkOutput>& /*network_output*/) override
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
    std::string getSearchInfo() const override
    {
        const Action a = getSearchAction();
        return "action: " + std::to_string(a.getActionID()) + " player: " + std::to_string(static_cast<int>(a.getPlayer())) + (isResign() ? " (resign)" : "");
    }
    void setNetwork(const std::shared_ptr<network::Network>& network) override // ref zero_actor.cpp:100-112
    {
        if (!network) { std::cerr << "setNetwork: null network" << std::endl; std::abort(); }
        if (worker_) { mz_worker_destroy(worker_); worker_ = nullptr; }
        std::string conf = config::mzgpuCollectConfiguration();
        // (n + 1) * action_size == tree_node_size (ref actor_group.cpp:183): the pool is sized from actor_num_simulation in the configuration
        conf += ":zero_num_parallel_games=1:mz_manual_step=true:nn_file_name=" + network->getNetworkFileName();
        worker_ = mz_worker_create(network->getGPUID(), conf.c_str(), nullptr, nullptr, 0);
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
    uint64_t tree_node_size_;
    int cycles_in_search_ = 0;
    bool fresh_ = false; // the worker has just been created: its game is new
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
