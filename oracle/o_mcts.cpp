// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  MCTS + Gumbel root logic.
// PARITY UNPINNED (actor/mcts.cpp cannot be compiled here without Boost stand-ins); every
// function restates the cited reference lines with the same operation order and the same
// float/double promotions (compile with -ffp-contract=off -O2, no -ffast-math).
#include "oracle.h"
#include <algorithm>
#include <cassert>
#include <cmath>
#include <limits>
#include <sstream>

namespace mzo {

void MCTSNode::reset() // ref mcts.cpp:5-18
{
    num_children_ = 0;
    hidden_state_data_index_ = -1;
    mean_ = 0.0f;
    count_ = 0.0f;
    virtual_loss_ = 0.0f;
    policy_ = 0.0f;
    policy_logit_ = 0.0f;
    policy_noise_ = 0.0f;
    value_ = 0.0f;
    reward_ = 0.0f;
    first_child_ = -1;
}

void MCTSNode::add(float value, float weight) // ref mcts.cpp:20-28
{
    if (count_ + weight <= 0) {
        reset();
    } else {
        count_ += weight;
        mean_ += weight * (value - mean_) / count_;
    }
}

void MCTS::reset() // ref tree.h:64-69, mcts.cpp:77-82
{
    if (nodes_.empty()) { nodes_.resize(1 + tree_node_size_); }
    current_node_size_ = 1;
    nodes_[0].reset();
    hidden_.clear();
    tree_value_bound_.clear();
}

float MCTS::getNormalizedMean(const MCTSNode* n) const // ref mcts.cpp:40-53
{
    float value = n->reward_ + cfg_->actor_mcts_reward_discount * n->mean_;
    if (cfg_->actor_mcts_value_rescale) {
        if (tree_value_bound_.size() < 2) { return 1.0f; }
        const float value_lower_bound = tree_value_bound_.begin()->first;
        const float value_upper_bound = tree_value_bound_.rbegin()->first;
        value = (value - value_lower_bound) / (value_upper_bound - value_lower_bound);
        value = ::fmin(static_cast<double>(1), ::fmax(static_cast<double>(-1), static_cast<double>(2 * value - 1))); // double libm, exact here
    }
    value = (n->action_.getPlayer() == charToPlayer(cfg_->actor_mcts_value_flipping_player) ? -value : value);
    value = (value * n->count_ - n->virtual_loss_) / n->getCountWithVirtualLoss();
    return value;
}

float MCTS::getNormalizedPUCTScore(const MCTSNode* n, int total_simulation, float init_q_value) const // ref mcts.cpp:55-61
{
    // (1 + N + base) / base is float; log() is the double libm function; init + log(..) is double -> float.
    float puct_bias = cfg_->actor_mcts_puct_init + ::log(static_cast<double>((1 + total_simulation + cfg_->actor_mcts_puct_base) / cfg_->actor_mcts_puct_base));
    // bias * P is float; * sqrt(int) promotes to double; / (1 + count) [float] in double; -> float.
    float value_u = (puct_bias * n->policy_ * ::sqrt(static_cast<double>(total_simulation))) / (1 + n->getCountWithVirtualLoss());
    float value_q = (n->getCountWithVirtualLoss() == 0 ? init_q_value : getNormalizedMean(n));
    return value_u + value_q;
}

bool MCTS::isResign(const MCTSNode* selected) const // ref mcts.cpp:84-89
{
    float root_win_rate = getNormalizedMean(root());
    float action_win_rate = getNormalizedMean(selected);
    return (-root_win_rate < cfg_->actor_resign_threshold && action_win_rate < cfg_->actor_resign_threshold);
}

MCTSNode* MCTS::selectChildByMaxCount(const MCTSNode* node) // ref mcts.cpp:91-104
{
    float max_count = 0.0f;
    MCTSNode* selected = nullptr;
    for (int i = 0; i < node->num_children_; ++i) {
        MCTSNode* c = child(node, i);
        if (c->count_ <= max_count) { continue; }
        max_count = c->count_;
        selected = c;
    }
    return selected;
}

MCTSNode* MCTS::selectChildBySoftmaxCount(const MCTSNode* node, float temperature, float value_threshold) // ref mcts.cpp:106-124
{
    MCTSNode* selected = nullptr;
    MCTSNode* best_child = selectChildByMaxCount(node);
    float best_mean = getNormalizedMean(best_child);
    float sum = 0.0f;
    for (int i = 0; i < node->num_children_; ++i) {
        MCTSNode* c = child(node, i);
        float count = std::pow(c->count_, 1 / temperature);
        float mean = getNormalizedMean(c);
        if (count == 0 || (mean < best_mean - value_threshold)) { continue; }
        sum += count;
        float rand = rng_->randReal(sum);
        if (selected == nullptr || rand < count) { selected = c; }
    }
    return selected;
}

std::string MCTS::getSearchDistributionString() const // ref mcts.cpp:126-137
{
    const MCTSNode* r = root();
    std::ostringstream oss;
    for (int i = 0; i < r->num_children_; ++i) {
        const MCTSNode* c = child(r, i);
        if (c->count_ == 0) { continue; }
        oss << (oss.str().empty() ? "" : ",") << c->action_.getActionID() << ":" << c->count_;
    }
    return oss.str();
}

std::vector<MCTSNode*> MCTS::selectFromNode(MCTSNode* start) // ref mcts.cpp:139-149
{
    MCTSNode* node = start;
    std::vector<MCTSNode*> node_path{node};
    while (!node->isLeaf()) {
        node = selectChildByPUCTScore(node);
        node_path.push_back(node);
    }
    return node_path;
}

void MCTS::expand(MCTSNode* leaf, const std::vector<ActionCandidate>& cands) // ref mcts.cpp:151-164, tree.h:71-77
{
    assert(current_node_size_ + cands.size() <= 1 + tree_node_size_);
    leaf->first_child_ = static_cast<int>(current_node_size_);
    current_node_size_ += cands.size();
    leaf->num_children_ = static_cast<int>(cands.size());
    for (size_t i = 0; i < cands.size(); ++i) {
        MCTSNode* c = child(leaf, static_cast<int>(i));
        c->reset();
        c->action_ = cands[i].action_;
        c->policy_ = cands[i].policy_;
        c->policy_logit_ = cands[i].policy_logit_;
    }
}

void MCTS::backup(const std::vector<MCTSNode*>& path, float value, float reward) // ref mcts.cpp:166-179
{
    float updated_value = value;
    path.back()->value_ = value;
    path.back()->reward_ = reward;
    for (int i = static_cast<int>(path.size() - 1); i >= 0; --i) {
        MCTSNode* node = path[i];
        float old_mean = node->reward_ + cfg_->actor_mcts_reward_discount * node->mean_;
        node->add(updated_value);
        updateTreeValueBound(old_mean, node->reward_ + cfg_->actor_mcts_reward_discount * node->mean_);
        updated_value = node->reward_ + cfg_->actor_mcts_reward_discount * updated_value;
    }
}

MCTSNode* MCTS::selectChildByPUCTScore(const MCTSNode* node) // ref mcts.cpp:181-198
{
    MCTSNode* selected = nullptr;
    int total_simulation = node->getCountWithVirtualLoss() - 1;
    float init_q_value = calculateInitQValue(node);
    float best_score = std::numeric_limits<float>::lowest(), best_policy = std::numeric_limits<float>::lowest();
    for (int i = 0; i < node->num_children_; ++i) {
        MCTSNode* c = child(node, i);
        float score = getNormalizedPUCTScore(c, total_simulation, init_q_value);
        if (score < best_score || (score == best_score && c->policy_ <= best_policy)) { continue; }
        best_score = score;
        best_policy = c->policy_;
        selected = c;
    }
    return selected;
}

float MCTS::calculateInitQValue(const MCTSNode* node) const // ref mcts.cpp:200-217
{
    float sum_of_win = 0.0f, sum = 0.0f;
    for (int i = 0; i < node->num_children_; ++i) {
        const MCTSNode* c = child(node, i);
        if (c->getCountWithVirtualLoss() == 0) { continue; }
        sum_of_win += getNormalizedMean(c);
        sum += 1;
    }
    if (cfg_->atari_init_q) { return (sum > 0 ? sum_of_win / sum : 1.0f); } // #if ATARI, mcts.cpp:211-213
    return (sum_of_win - 1) / (sum + 1);
}

void MCTS::updateTreeValueBound(float old_value, float new_value) // ref mcts.cpp:219-228
{
    if (!cfg_->actor_mcts_value_rescale) { return; }
    if (tree_value_bound_.count(old_value)) {
        --tree_value_bound_[old_value];
        if (tree_value_bound_[old_value] == 0) { tree_value_bound_.erase(old_value); }
    }
    ++tree_value_bound_[new_value];
}

// -------------------------------------------------------------------------------------
// Gumbel — ref actor/gumbel_zero.cpp
// -------------------------------------------------------------------------------------
std::string GumbelZero::getMCTSPolicy(const Config& cfg, MCTS& mcts) const // ref gumbel_zero.cpp:9-58
{
    float pi_sum = 0.0f, q_sum = 0.0f;
    const MCTSNode* root = mcts.root();
    for (int i = 0; i < root->num_children_; ++i) {
        const MCTSNode* c = mcts.child(root, i);
        if (c->count_ == 0) { continue; }
        float value = mcts.getNormalizedMean(c);
        pi_sum += c->policy_;
        q_sum += c->policy_ * value;
    }
    float value_pi = root->value_;
    if (cfg.actor_mcts_value_rescale) {
        if (mcts.tree_value_bound_.size() < 2) {
            value_pi = 1.0f;
        } else {
            const float lo = mcts.tree_value_bound_.begin()->first;
            const float hi = mcts.tree_value_bound_.rbegin()->first;
            value_pi = (value_pi - lo) / (hi - lo);
            value_pi = ::fmin(static_cast<double>(1), ::fmax(static_cast<double>(-1), static_cast<double>(2 * value_pi - 1)));
        }
    }
    value_pi = (mcts.child(root, 0)->action_.getPlayer() == charToPlayer(cfg.actor_mcts_value_flipping_player) ? -value_pi : value_pi);
    float non_visited_node_value = 1.0 / (1 + cfg.actor_num_simulation) * (value_pi + (cfg.actor_num_simulation / pi_sum) * q_sum);

    std::unordered_map<int, float> new_logits;
    float max_logit = -std::numeric_limits<float>::max();
    float max_child_count = 0;
    for (int i = 0; i < root->num_children_; ++i) { max_child_count = ::fmax(static_cast<double>(max_child_count), static_cast<double>(mcts.child(root, i)->count_)); }
    for (int i = 0; i < root->num_children_; ++i) {
        const MCTSNode* c = mcts.child(root, i);
        float value = (c->count_ == 0 ? non_visited_node_value : mcts.getNormalizedMean(c));
        float logit_without_noise = c->policy_logit_ - c->policy_noise_;
        float score = logit_without_noise + (cfg.actor_gumbel_sigma_visit_c + max_child_count) * cfg.actor_gumbel_sigma_scale_c * value;
        new_logits.insert({c->action_.getActionID(), score});
        max_logit = ::fmax(static_cast<double>(max_logit), static_cast<double>(score));
    }
    std::ostringstream oss;
    for (auto& logit : new_logits) { // libstdc++ hashtable iteration order is part of the record format
        logit.second = logit.second - max_logit;
        if (logit.second < -38) { continue; }
        oss << (oss.str().empty() ? "" : ",") << logit.first << ":" << ::exp(static_cast<double>(logit.second)); // double exp, ostream 6 significant digits
    }
    return oss.str();
}

MCTSNode* GumbelZero::decideActionNode(const Config& cfg, MCTS& mcts) // ref gumbel_zero.cpp:60-72
{
    if (cfg.actor_select_action_by_count) {
        sortCandidatesByScore(cfg, mcts);
        return candidates_[0];
    } else if (cfg.actor_select_action_by_softmax_count) {
        return mcts.selectChildBySoftmaxCount(mcts.root(), cfg.actor_select_action_softmax_temperature);
    }
    return nullptr;
}

std::vector<MCTSNode*> GumbelZero::selection(MCTS& mcts) // ref gumbel_zero.cpp:74-88
{
    std::vector<MCTSNode*> node_path;
    if (mcts.getNumSimulation() == 0) {
        node_path = mcts.select();
    } else {
        std::sort(candidates_.begin(), candidates_.end(), [](const MCTSNode* lhs, const MCTSNode* rhs) {
            return (lhs->count_ < rhs->count_ || (lhs->count_ == rhs->count_ && lhs->policy_logit_ > rhs->policy_logit_));
        });
        node_path = mcts.selectFromNode(candidates_[0]);
        node_path.insert(node_path.begin(), mcts.root());
    }
    return node_path;
}

void GumbelZero::sequentialHalving(const Config& cfg, MCTS& mcts) // ref gumbel_zero.cpp:90-119
{
    if (mcts.getNumSimulation() == 1) {
        candidates_.clear();
        for (int i = 0; i < mcts.root()->num_children_; ++i) { candidates_.push_back(mcts.child(mcts.root(), i)); }
        std::sort(candidates_.begin(), candidates_.end(), [](const MCTSNode* lhs, const MCTSNode* rhs) { return lhs->policy_logit_ > rhs->policy_logit_; });
        if (static_cast<int>(candidates_.size()) > cfg.actor_gumbel_sample_size) { candidates_.resize(cfg.actor_gumbel_sample_size); }
        sample_size_ = cfg.actor_gumbel_sample_size;
        simulation_budget_ = std::max(1.0, std::floor(cfg.actor_num_simulation / (std::log2(cfg.actor_gumbel_sample_size) * sample_size_)));
    } else {
        bool all_candidates_reach_budget = true;
        for (auto node : candidates_) {
            if (node->count_ >= simulation_budget_) { continue; }
            all_candidates_reach_budget = false;
            break;
        }
        if (all_candidates_reach_budget) {
            int next_budget = std::floor(cfg.actor_num_simulation / (std::log2(cfg.actor_gumbel_sample_size) * sample_size_ / 2));
            if (next_budget > 0 && sample_size_ > 2) {
                sample_size_ /= 2;
                sortCandidatesByScore(cfg, mcts);
                if (static_cast<int>(candidates_.size()) > sample_size_) { candidates_.resize(sample_size_); }
                simulation_budget_ = candidates_[0]->count_ + next_budget;
            }
        }
    }
}

void GumbelZero::sortCandidatesByScore(const Config& cfg, MCTS& mcts) // ref gumbel_zero.cpp:121-137
{
    float max_child_count = 0;
    for (int i = 0; i < mcts.root()->num_children_; ++i) { max_child_count = ::fmax(static_cast<double>(max_child_count), static_cast<double>(mcts.child(mcts.root(), i)->count_)); }
    std::sort(candidates_.begin(), candidates_.end(), [&](const MCTSNode* lhs, const MCTSNode* rhs) {
        float min_value = -std::numeric_limits<float>::max();
        float lhs_value = mcts.getNormalizedMean(lhs);
        float lhs_score = lhs->policy_logit_ + (cfg.actor_gumbel_sigma_visit_c + max_child_count) * cfg.actor_gumbel_sigma_scale_c * lhs_value;
        lhs_score = (lhs->count_ > 0 ? lhs_score : min_value);
        float rhs_value = mcts.getNormalizedMean(rhs);
        float rhs_score = rhs->policy_logit_ + (cfg.actor_gumbel_sigma_visit_c + max_child_count) * cfg.actor_gumbel_sigma_scale_c * rhs_value;
        rhs_score = (rhs->count_ > 0 ? rhs_score : min_value);
        return lhs_score > rhs_score;
    });
}

} // namespace mzo
