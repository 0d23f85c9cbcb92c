// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Per-game actor + lock-step group loop + records.
// PARITY UNPINNED (actor/zero_actor.cpp needs utils/time_system.h -> Boost; actor_group.cpp needs
// utils/paralleler.h -> boost/thread).  Restated line by line from the cited reference code.
#include "o_actor.h"
#include <algorithm>
#include <cassert>
#include <sstream>
#include <thread>

namespace mzo {

// ---------------------------------------------------------------------------------------------
// ZeroActor — ref actor/zero_actor.cpp, actor/base_actor.cpp
// ---------------------------------------------------------------------------------------------
ZeroActor::ZeroActor(const Config* cfg, Random* rng, const NetDesc* nd, NetQueue* q, uint64_t tree_node_size)
    : cfg_(cfg), rng_(rng), nd_(nd), q_(q), env_(createEnv(*cfg, rng)), mcts_(cfg, rng, tree_node_size)
{
}

void ZeroActor::reset() // ref zero_actor.cpp:23-27, base_actor.cpp:8-13
{
    env_->reset();
    action_info_history_.clear();
    resetSearch();
    enable_resign_ = (rng_->randReal() < cfg_->zero_disable_resign_ratio ? false : true);
}

void ZeroActor::resetSearch() // ref zero_actor.cpp:29-34, base_actor.cpp:15-20
{
    nn_evaluation_batch_id_ = -1;
    mcts_.reset();
    node_path_.clear();
    mcts_.root()->action_ = Action(-1, getPreviousPlayer(env_->getTurn(), env_->getNumPlayer()));
}

bool ZeroActor::act(const Action& a) // ref base_actor.cpp:22-30
{
    bool can_act = env_->act(a);
    if (can_act) {
        action_info_history_.resize(env_->getActionHistory().size());
        action_info_history_.back() = getActionInfo();
    }
    return can_act;
}

std::vector<std::pair<std::string, std::string>> ZeroActor::getActionInfo() const // ref zero_actor.cpp:114-119, base_actor.cpp:59-66
{
    if (!(mcts_.root()->count_ > 0)) { return {}; }
    std::vector<std::pair<std::string, std::string>> info;
    std::string p = (cfg_->actor_use_gumbel ? gumbel_zero_.getMCTSPolicy(*cfg_, const_cast<MCTS&>(mcts_)) : mcts_.getSearchDistributionString()); // zero_actor.h:50
    info.push_back({"P", p});
    info.push_back({"V", std::to_string(mcts_.root()->mean_)}); // zero_actor.h:51
    std::ostringstream oss;
    oss << env_->getReward(); // zero_actor.cpp:121-126
    info.push_back({"R", oss.str()});
    return info;
}

static std::string escapeSGFString(const std::string& str) // ref base_env.h:303-313
{
    std::string special = "()[]\\";
    std::string escaped;
    for (char c : str) {
        if (special.find(c) != std::string::npos) { escaped += '\\'; }
        escaped += c;
    }
    return escaped;
}

std::string ZeroActor::getRecord(const std::vector<std::pair<std::string, std::string>>& extra_tags) const
{
    // ref base_actor.cpp:39-57 + base_env.h:122-129,207-233,363-367 + go.h:129-133 (tag order = insertion order of VectorMap)
    std::vector<std::pair<std::string, std::string>> tags;
    auto addTag = [&](const std::string& k, const std::string& v) {
        for (auto& t : tags) { if (t.first == k) { t.second = v; return; } }
        tags.push_back({k, v});
    };
    addTag("GM", env_->name());
    addTag("RE", "0");
    addTag("RE", std::to_string(env_->getEvalScore()));
    { // ref base_env.h:216-220: OBS = compressString(all observation strings concatenated); board games have none -> ""
        std::string observations;
        for (const auto& obs : env_->getObservationHistory()) { observations += obs; }
        addTag("OBS", compressString(observations));
    }
    for (auto& t : env_->loaderTags()) { addTag(t.first, t.second); }
    // ref atari.cpp:187-197 AtariEnvLoader::loadFromEnvironment: action i gets L[lives] when the lives before it are fewer than before action i-1
    std::vector<std::pair<size_t, std::string>> lives_tags;
    {
        const std::vector<int> lives_history = env_->getLivesHistory();
        if (!lives_history.empty()) {
            int previous_lives = lives_history[0];
            for (size_t i = 0; i < env_->getActionHistory().size(); ++i) {
                const int lives = lives_history[i];
                if (lives < previous_lives) { lives_tags.push_back({i, std::to_string(lives)}); }
                previous_lives = lives;
            }
        }
    }
    addTag("EV", cfg_->nn_file_name.substr(cfg_->nn_file_name.find_last_of('/') + 1));
    if (!isEnvTerminal()) {
        float result = env_->getEvalScore(true);
        std::ostringstream oss;
        oss << result;
        addTag("RE", oss.str());
    }
    for (auto& t : extra_tags) { addTag(t.first, t.second); }

    std::ostringstream oss;
    oss << "(;";
    for (const auto& t : tags) { oss << t.first << "[" << escapeSGFString(t.second) << "]"; }
    const auto& actions = env_->getActionHistory();
    for (size_t i = 0; i < actions.size(); ++i) {
        oss << ";" << playerToChar(actions[i].getPlayer()) << "[" << actions[i].getActionID() << "]";
        bool has_l = false; // VectorMap semantics: an existing "L" key keeps its place and takes the new value
        std::string l_value;
        for (const auto& lt : lives_tags) { if (lt.first == i) { has_l = true; l_value = lt.second; } }
        if (action_info_history_.size() > i) {
            for (const auto& info : action_info_history_[i]) {
                const bool is_l = has_l && info.first == "L";
                oss << info.first << "[" << escapeSGFString(is_l ? l_value : info.second) << "]";
                if (is_l) { has_l = false; }
            }
        }
        if (has_l) { oss << "L[" << escapeSGFString(l_value) << "]"; }
    }
    oss << ")";
    return oss.str();
}

std::vector<MCTSNode*> ZeroActor::selection() // ref zero_actor.h:57
{
    return (cfg_->actor_use_gumbel ? gumbel_zero_.selection(mcts_) : mcts_.select());
}

std::unique_ptr<Env> ZeroActor::getEnvironmentTransition(const std::vector<MCTSNode*>& path) const // ref zero_actor.cpp:247-252
{
    std::unique_ptr<Env> env = env_->clone();
    for (size_t i = 1; i < path.size(); ++i) { env->act(path[i]->action_); }
    return env;
}

void ZeroActor::beforeNNEvaluation() // ref zero_actor.cpp:51-72
{
    node_path_ = selection();
    last_path_idx_.clear();
    for (auto* n : node_path_) { last_path_idx_.push_back(mcts_.indexOf(n)); }
    if (nd_->type == 0) {
        std::unique_ptr<Env> env_transition = getEnvironmentTransition(node_path_);
        feature_rotation_ = cfg_->actor_use_random_rotation_features ? static_cast<Rotation>(rng_->randInt() % static_cast<int>(kRotateSize)) : kRotationNone;
        nn_evaluation_batch_id_ = q_->pushBack(env_transition->getFeatures(feature_rotation_), slot_override_);
    } else {
        if (mcts_.getNumSimulation() == 0) {
            nn_evaluation_batch_id_ = q_->pushBackInitial(env_->getFeatures(), slot_override_);
        } else {
            MCTSNode* leaf = node_path_.back();
            MCTSNode* parent = node_path_[node_path_.size() - 2];
            const std::vector<float>& hidden = mcts_.hidden(parent->hidden_state_data_index_);
            nn_evaluation_batch_id_ = q_->pushBackRecurrent(hidden, env_->getActionFeatures(leaf->action_), slot_override_);
        }
    }
}

void ZeroActor::afterNNEvaluation(const NetOutput& out) // ref zero_actor.cpp:74-98
{
    MCTSNode* leaf = node_path_.back();
    last_cand_actions_.clear();
    if (nd_->type == 0) {
        std::unique_ptr<Env> env_transition = getEnvironmentTransition(node_path_);
        if (!env_transition->isTerminal()) {
            auto cands = calculateAlphaZeroActionPolicy(*env_transition, out, feature_rotation_);
            for (auto& c : cands) { last_cand_actions_.push_back(c.action_.getActionID()); }
            mcts_.expand(leaf, cands);
            mcts_.backup(node_path_, out.value_, env_transition->getReward());
        } else {
            mcts_.backup(node_path_, env_transition->getEvalScore(), env_transition->getReward());
        }
    } else {
        auto cands = calculateMuZeroActionPolicy(leaf, out);
        for (auto& c : cands) { last_cand_actions_.push_back(c.action_.getActionID()); }
        mcts_.expand(leaf, cands);
        mcts_.backup(node_path_, out.value_, out.reward_);
        leaf->hidden_state_data_index_ = mcts_.storeHidden(out.hidden_state_);
    }
    if (leaf == mcts_.root()) { addNoiseToNodeChildren(leaf); }
    if (isSearchDone()) { selected_node_ = decideActionNode(); } // handleSearchDone, zero_actor.cpp:159-176 (log text omitted)
    if (cfg_->actor_use_gumbel) { gumbel_zero_.sequentialHalving(*cfg_, mcts_); }
}

MCTSNode* ZeroActor::decideActionNode() // ref zero_actor.cpp:178-192
{
    if (cfg_->actor_use_gumbel) { return gumbel_zero_.decideActionNode(*cfg_, mcts_); }
    if (cfg_->actor_select_action_by_count) { return mcts_.selectChildByMaxCount(mcts_.root()); }
    if (cfg_->actor_select_action_by_softmax_count) { return mcts_.selectChildBySoftmaxCount(mcts_.root(), cfg_->actor_select_action_softmax_temperature); }
    return nullptr;
}

void ZeroActor::addNoiseToNodeChildren(MCTSNode* node) // ref zero_actor.cpp:194-213
{
    if (cfg_->actor_use_dirichlet_noise) {
        const float epsilon = cfg_->actor_dirichlet_noise_epsilon;
        std::vector<float> dirichlet_noise = rng_->randDirichlet(cfg_->actor_dirichlet_noise_alpha, node->num_children_);
        for (int i = 0; i < node->num_children_; ++i) {
            MCTSNode* c = mcts_.child(node, i);
            c->policy_noise_ = dirichlet_noise[i];
            c->policy_ = (1 - epsilon) * c->policy_ + epsilon * dirichlet_noise[i];
        }
    } else if (cfg_->actor_use_gumbel_noise) {
        std::vector<float> gumbel_noise = rng_->randGumbel(node->num_children_);
        for (int i = 0; i < node->num_children_; ++i) {
            MCTSNode* c = mcts_.child(node, i);
            c->policy_noise_ = gumbel_noise[i];
            c->policy_logit_ = c->policy_logit_ + gumbel_noise[i];
        }
    }
}

std::vector<ActionCandidate> ZeroActor::calculateAlphaZeroActionPolicy(const Env& env_transition, const NetOutput& out, Rotation rot) // ref zero_actor.cpp:215-229
{
    std::vector<ActionCandidate> cands;
    for (size_t action_id = 0; action_id < out.policy_.size(); ++action_id) {
        Action action(static_cast<int>(action_id), env_transition.getTurn());
        if (!env_transition.isLegalAction(action)) { continue; }
        int rotated_id = env_transition.getRotateAction(static_cast<int>(action_id), rot);
        cands.push_back(ActionCandidate(action, out.policy_[rotated_id], out.policy_logits_[rotated_id]));
    }
    std::sort(cands.begin(), cands.end(), [](const ActionCandidate& lhs, const ActionCandidate& rhs) { return lhs.policy_ > rhs.policy_; });
    return cands;
}

std::vector<ActionCandidate> ZeroActor::calculateMuZeroActionPolicy(MCTSNode* leaf, const NetOutput& out) // ref zero_actor.cpp:231-245
{
    std::vector<ActionCandidate> cands;
    Player turn = getNextPlayer(leaf->action_.getPlayer(), env_->getNumPlayer());
    for (size_t action_id = 0; action_id < out.policy_.size(); ++action_id) {
        const Action action(static_cast<int>(action_id), turn);
        if (leaf == mcts_.root() && !env_->isLegalAction(action)) { continue; }
        cands.push_back(ActionCandidate(action, out.policy_[action_id], out.policy_logits_[action_id]));
    }
    std::sort(cands.begin(), cands.end(), [](const ActionCandidate& lhs, const ActionCandidate& rhs) { return lhs.policy_ > rhs.policy_; });
    return cands;
}

// ---------------------------------------------------------------------------------------------
// NetQueue — ref network/alphazero_network.h:48-104, muzero_network.h:64-178 (queue + per-sample outputs)
// ---------------------------------------------------------------------------------------------
std::vector<NetOutput> NetQueue::run()
{
    std::vector<NetOutput> outs;
    const NetDesc& d = net->desc;
    const int A = d.action_size, hs = d.num_hidden_channels * d.hidden_channel_height * d.hidden_channel_width;
    int B = 0;
    std::vector<float> policy, logit, value, reward, hidden;
    if (!az.empty()) {
        B = static_cast<int>(az.size() / feat_size);
        policy.resize(size_t(B) * A); logit.resize(size_t(B) * A); value.resize(B);
        net->forwardAZ(az.data(), B, policy.data(), logit.data(), value.data());
        az.clear();
    } else if (!init.empty()) {
        B = static_cast<int>(init.size() / feat_size);
        policy.resize(size_t(B) * A); logit.resize(size_t(B) * A); value.resize(B); hidden.resize(size_t(B) * hs);
        net->initialMZ(init.data(), B, policy.data(), logit.data(), value.data(), hidden.data());
        init.clear();
    } else if (!rec_h.empty()) {
        B = static_cast<int>(rec_h.size() / hs);
        policy.resize(size_t(B) * A); logit.resize(size_t(B) * A); value.resize(B); hidden.resize(size_t(B) * hs); reward.assign(B, 0.0f);
        net->recurrentMZ(rec_h.data(), rec_a.data(), B, policy.data(), logit.data(), value.data(), reward.data(), hidden.data());
        rec_h.clear();
        rec_a.clear();
    }
    leaf_evals += B;
    outs.resize(B);
    for (int i = 0; i < B; ++i) {
        outs[i].value_ = value[i];
        outs[i].reward_ = reward.empty() ? 0.0f : reward[i];
        outs[i].policy_.assign(policy.begin() + size_t(i) * A, policy.begin() + size_t(i + 1) * A);
        outs[i].policy_logits_.assign(logit.begin() + size_t(i) * A, logit.begin() + size_t(i + 1) * A);
        if (!hidden.empty()) { outs[i].hidden_state_.assign(hidden.begin() + size_t(i) * hs, hidden.begin() + size_t(i + 1) * hs); }
    }
    return outs;
}

// ---------------------------------------------------------------------------------------------
// Group — ref actor/actor_group.cpp (one slave thread == the deterministic contract)
// ---------------------------------------------------------------------------------------------
thread_local int Group::cur_thread_ = -1;
Group::Group(const Config& cfg, const NetDesc& nd, const float* raw, size_t nraw) : cfg_(cfg), nd_(nd)
{
    net_ = Net::create(nd, raw, nraw);
    q_.net = net_.get();
    q_.feat_size = size_t(nd.num_input_channels) * nd.input_channel_height * nd.input_channel_width;
    // ref mode_handler.cpp:62 main-thread seed, used by createActors (actor_group.cpp:179-187 -> reset() -> resign coin)
    main_rng_.seed(cfg_.program_seed);
    uint64_t tree_node_size = static_cast<uint64_t>(cfg_.actor_num_simulation + 1) * nd.action_size; // actor_group.cpp:183
    for (int i = 0; i < cfg_.zero_num_parallel_games; ++i) {
        actors_.emplace_back(std::make_unique<ZeroActor>(&cfg_, &main_rng_, &nd_, &q_, tree_node_size));
        actors_.back()->reset();
    }
    // ref actor_group.cpp:66-70: slave thread 0 seeds ITS generator with program_seed + 0
    slave_rng_.seed(cfg_.program_seed + 0);
    for (auto& a : actors_) { a->rng_ = &slave_rng_; a->mcts_.rng_ = &slave_rng_; a->env_->rng_ = &slave_rng_; }
    // T slave threads (oracle_throughput_threads = the reference's zero_num_threads > 1): thread t seeds ITS generator with program_seed + t (actor_group.cpp:66-70).
    // The reference hands actors to threads first-come-first-served (:18-22, utils/paralleler.h) — a race: which thread's generator an actor draws from differs
    // from run to run.  Here the partition is STATIC, one of the schedules that race can produce: actor i belongs to thread i * T / B (contiguous blocks, every
    // thread takes its actors in index order), and the lines of a cycle are emitted in thread order = actor order.  The worker's host section follows the same
    // partition (worker.cpp streamOf), so records stay comparable for every T.
    if (cfg_.oracle_throughput_threads > 1) {
        const size_t T = static_cast<size_t>(cfg_.oracle_throughput_threads), B = actors_.size();
        for (size_t t = 0; t < T; ++t) {
            thread_rngs_.emplace_back(std::make_unique<Random>());
            thread_rngs_.back()->seed(cfg_.program_seed + static_cast<int>(t));
        }
        thread_lines_.resize(T);
        thread_of_.resize(B);
        for (size_t i = 0; i < B; ++i) {
            thread_of_[i] = static_cast<int>(i * T / B);
            Random* r = thread_rngs_[thread_of_[i]].get();
            actors_[i]->rng_ = r; actors_[i]->mcts_.rng_ = r; actors_[i]->env_->rng_ = r;
        }
    }
}

std::pair<int, int> Group::calculateTrainingDataRange(const ZeroActor& actor) const // ref actor_group.cpp:52-64
{
    int game_length = static_cast<int>(actor.env_->getActionHistory().size());
    int data_start = 0, data_end = game_length - 1;
    const int seq = cfg_.zero_actor_intermediate_sequence_length;
    if (seq > 0) {
        const int un = cfg_.learner_muzero_unrolling_step + cfg_.learner_n_step_return;
        data_end = std::max(0, (actor.env_->isTerminal() ? data_end : data_end - un));
        data_start = std::max(0, (actor.env_->isTerminal() ? data_end - data_end % seq : data_end + 1 - seq));
        if (actor.env_->isTerminal() && (data_end % seq < un)) { data_start = std::max(0, data_start - seq); }
    }
    return {data_start, data_end};
}

void Group::outputGame(ZeroActor& actor) // ref actor_group.cpp:24-50
{
    int game_length = static_cast<int>(actor.env_->getActionHistory().size());
    std::pair<int, int> data_range = calculateTrainingDataRange(actor);
    std::ostringstream oss;
    bool is_terminal = (cfg_.zero_actor_intermediate_sequence_length == 0 || actor.isEnvTerminal());
    oss << "SelfPlay " << (is_terminal ? "true" : "false") << " " << (data_range.second - data_range.first + 1) << " " << game_length << " "
        << actor.env_->getEvalScore(!actor.isEnvTerminal()) << " "
        << actor.getRecord({{"DLEN", std::to_string(data_range.first) + "-" + std::to_string(data_range.second)}}) << " "
        << "#";
    if (!is_terminal) {
        // (a resignation before the first move of an intermediate-sequence game — game_length 0, range 0-0 — indexes an EMPTY history in the reference:
        // undefined behaviour there, found by the fuzz sweep; the restatement and the worker skip what does not exist)
        for (int i = data_range.first; i <= data_range.second && i < static_cast<int>(actor.action_info_history_.size()); ++i) { actor.action_info_history_[i].clear(); }
    }
    std::lock_guard<std::mutex> lock(out_mutex_);
    if (cur_thread_ >= 0) { thread_lines_[cur_thread_].push_back(oss.str()); } // (T threads: merged in thread order at the end of the cycle)
    else { lines_.push_back(oss.str()); }
    if (is_terminal) { ++games_; }
}

void Group::handleSearchDone(int actor_id) // ref actor_group.cpp:116-134
{
    ZeroActor& actor = *actors_[actor_id];
    if (!actor.isResign()) { actor.act(actor.getSearchAction()); }
    bool is_endgame = (actor.isResign() || actor.isEnvTerminal());
    if (is_endgame) {
        outputGame(actor);
        actor.reset();
    } else {
        int game_length = static_cast<int>(actor.env_->getActionHistory().size());
        int seq = cfg_.zero_actor_intermediate_sequence_length;
        if (seq > 0 && game_length >= seq && (game_length - cfg_.learner_n_step_return - cfg_.learner_muzero_unrolling_step) % seq == 0) { outputGame(actor); }
        actor.resetSearch();
    }
}

void Group::cycle() // ref actor_group.cpp:81-114 (one CPU phase + one GPU phase)
{
    if (!thread_rngs_.empty()) { // T slave threads with a static partition (constructor); batch slots are assigned up front in actor order
        const int T = static_cast<int>(thread_rngs_.size()), B = static_cast<int>(actors_.size());
        std::vector<std::thread> th;
        q_.reserveSlots(B, nd_);
        for (int t = 0; t < T; ++t) {
            th.emplace_back([this, t, B]() {
                cur_thread_ = t;
                for (int i = 0; i < B; ++i) {
                    if (thread_of_[i] != t) { continue; }
                    ZeroActor& actor = *actors_[i];
                    const int out_id = actor.nn_evaluation_batch_id_;
                    if (out_id >= 0) {
                        actor.afterNNEvaluation(outputs_[out_id]);
                        if (actor.isSearchDone()) { handleSearchDone(i); }
                    }
                    actor.slot_override_ = i;
                    actor.beforeNNEvaluation();
                }
                cur_thread_ = -1;
            });
        }
        for (auto& t : th) { t.join(); }
        for (auto& tl : thread_lines_) { for (auto& l : tl) { lines_.push_back(std::move(l)); } tl.clear(); }
        outputs_ = q_.run();
        ++cycles_;
        return;
    }
    for (size_t i = 0; i < actors_.size(); ++i) { // doCPUJob in actor-index order
        ZeroActor& actor = *actors_[i];
        int out_id = actor.nn_evaluation_batch_id_;
        if (out_id >= 0) {
            actor.afterNNEvaluation(outputs_[out_id]);
            if (trace_) { traceAfter(static_cast<int>(i)); }
            if (actor.isSearchDone()) { handleSearchDone(static_cast<int>(i)); }
        }
        actor.beforeNNEvaluation();
        if (trace_) { traceBefore(static_cast<int>(i)); }
    }
    outputs_ = q_.run(); // doGPUJob
    ++cycles_;
}

int Group::command(const std::string& line, const float* raw, size_t nraw) // ref actor_group.cpp:200-252
{
    const std::string prefix = line.substr(0, line.find(' '));
    { // zero_actor_ignored_command (:204-212; default "reset_actors", configuration.cpp:47)
        std::istringstream ign(cfg_.zero_actor_ignored_command);
        std::string tok;
        while (ign >> tok) { if (tok == prefix) { return 0; } }
    }
    if (prefix == "reset_actors") { // :222-225 — every actor's reset() on the MAIN thread, i.e. with the main thread's generator (utils/random.h:38 thread_local)
        for (auto& a : actors_) {
            Random* own = a->rng_;
            a->rng_ = &main_rng_; a->mcts_.rng_ = &main_rng_; a->env_->rng_ = &main_rng_;
            a->reset();
            a->rng_ = own; a->mcts_.rng_ = own; a->env_->rng_ = own;
        }
        // do_cpu_job_ = true: the next cycle starts with a CPU phase — every cycle() does; nothing of the last GPU phase is consumed (batch ids are -1)
    } else if (prefix == "load_model") { // :226-232 — config::nn_file_name = args[1]; every network re-reads the file
        if (line.find(' ') == std::string::npos || !raw) { return -1; }
        std::unique_ptr<Net> nn = Net::create(nd_, raw, nraw);
        if (!nn) { return -1; }
        cfg_.nn_file_name = line.substr(line.find(' ') + 1);
        net_ = std::move(nn);
        q_.net = net_.get();
    } else if (prefix == "update_config") { // :233-241
        if (line.find(' ') == std::string::npos || !cfg_.loadFromString(line.substr(line.find(' ') + 1))) { return -1; }
    } else if (prefix == "start") { running_ = true; }
    else if (prefix == "stop") { running_ = false; }
    else if (prefix == "quit") { return 1; }
    return 0; // anything else (keep_alive, ...) falls through handleCommand's chain
}

void Group::traceBefore(int i)
{
    ZeroActor& a = *actors_[i];
    std::ostringstream oss;
    oss << "S " << cycles_ << " " << i << " rot=" << static_cast<int>(a.feature_rotation_) << " path=";
    for (size_t k = 0; k < a.last_path_idx_.size(); ++k) { oss << (k ? "," : "") << a.last_path_idx_[k]; }
    trace_lines_.push_back(oss.str());
}
void Group::traceAfter(int i)
{
    ZeroActor& a = *actors_[i];
    std::ostringstream oss;
    oss << "E " << cycles_ << " " << i << " cand=";
    for (size_t k = 0; k < a.last_cand_actions_.size(); ++k) { oss << (k ? "," : "") << a.last_cand_actions_[k]; }
    trace_lines_.push_back(oss.str());
    if (a.isSearchDone()) { // root child visit counts at the end of a search
        std::ostringstream r;
        r << "R " << cycles_ << " " << i << " counts=";
        const MCTSNode* root = a.mcts_.root();
        for (int k = 0; k < root->num_children_; ++k) { r << (k ? "," : "") << a.mcts_.child(root, k)->count_; }
        trace_lines_.push_back(r.str());
    }
}

} // namespace mzo
