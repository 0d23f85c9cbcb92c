// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  C entry points for ctypes (tests/, smoke(), bench cpu_baseline).
#include "o_actor.h"
#include <cstring>
#include <sstream>

using namespace mzo;

namespace {
struct Tree {
    Config cfg;
    Random rng;
    std::unique_ptr<MCTS> mcts;
    std::vector<MCTSNode*> path;
};
int copyOut(const std::string& s, char* buf, int cap)
{
    int n = static_cast<int>(s.size());
    if (buf && cap > 0) {
        int m = n < cap - 1 ? n : cap - 1;
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return n;
}
} // namespace

extern "C" {

// ---- math / rng / rotation / config ----
void mzo_expf(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) { y[i] = mz_expf(x[i]); } }
void mzo_tanhf(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) { y[i] = mz_tanhf(x[i]); } }

// kind: 0 randInt, 1 randReal, 2 randDirichlet(alpha, k) repeated, 3 randGumbel(k) repeated; out has n doubles
void mzo_rng_vector(int seed, int kind, int n, int k, float alpha, double* out)
{
    Random r;
    r.seed(seed);
    int i = 0;
    while (i < n) {
        if (kind == 0) { out[i++] = r.randInt(); }
        else if (kind == 1) { out[i++] = r.randReal(); }
        else {
            std::vector<float> v = (kind == 2 ? r.randDirichlet(alpha, k) : r.randGumbel(k));
            for (int j = 0; j < k && i < n; ++j) { out[i++] = v[j]; }
        }
    }
}
int mzo_rotate(int rotation, int pos, int board_size) { return getPositionByRotating(static_cast<Rotation>(rotation), pos, board_size); }
int mzo_reversed_rotation(int rotation) { return reversed_rotation[rotation]; }

int mzo_config_dump(const char* conf, char* buf, int cap)
{
    Config c;
    if (conf && *conf && !c.loadFromString(conf)) { return -1; }
    std::ostringstream o;
    o << "program_seed=" << c.program_seed << "\nprogram_auto_seed=" << c.program_auto_seed << "\nprogram_quiet=" << c.program_quiet
      << "\nactor_num_simulation=" << c.actor_num_simulation << "\nactor_mcts_puct_base=" << c.actor_mcts_puct_base
      << "\nactor_mcts_puct_init=" << c.actor_mcts_puct_init << "\nactor_mcts_reward_discount=" << c.actor_mcts_reward_discount
      << "\nactor_mcts_think_batch_size=" << c.actor_mcts_think_batch_size << "\nactor_mcts_think_time_limit=" << c.actor_mcts_think_time_limit
      << "\nactor_mcts_value_rescale=" << c.actor_mcts_value_rescale << "\nactor_mcts_value_flipping_player=" << c.actor_mcts_value_flipping_player
      << "\nactor_select_action_by_count=" << c.actor_select_action_by_count
      << "\nactor_select_action_by_softmax_count=" << c.actor_select_action_by_softmax_count
      << "\nactor_select_action_softmax_temperature=" << c.actor_select_action_softmax_temperature
      << "\nactor_select_action_softmax_temperature_decay=" << c.actor_select_action_softmax_temperature_decay
      << "\nactor_use_random_rotation_features=" << c.actor_use_random_rotation_features
      << "\nactor_use_dirichlet_noise=" << c.actor_use_dirichlet_noise << "\nactor_dirichlet_noise_alpha=" << c.actor_dirichlet_noise_alpha
      << "\nactor_dirichlet_noise_epsilon=" << c.actor_dirichlet_noise_epsilon << "\nactor_use_gumbel=" << c.actor_use_gumbel
      << "\nactor_use_gumbel_noise=" << c.actor_use_gumbel_noise << "\nactor_gumbel_sample_size=" << c.actor_gumbel_sample_size
      << "\nactor_gumbel_sigma_visit_c=" << c.actor_gumbel_sigma_visit_c << "\nactor_gumbel_sigma_scale_c=" << c.actor_gumbel_sigma_scale_c
      << "\nactor_resign_threshold=" << c.actor_resign_threshold << "\nzero_num_threads=" << c.zero_num_threads
      << "\nzero_num_parallel_games=" << c.zero_num_parallel_games << "\nzero_disable_resign_ratio=" << c.zero_disable_resign_ratio
      << "\nzero_actor_intermediate_sequence_length=" << c.zero_actor_intermediate_sequence_length
      << "\nzero_actor_ignored_command=" << c.zero_actor_ignored_command << "\nlearner_muzero_unrolling_step=" << c.learner_muzero_unrolling_step
      << "\nlearner_n_step_return=" << c.learner_n_step_return << "\nnn_file_name=" << c.nn_file_name << "\nnn_type_name=" << c.nn_type_name
      << "\nenv_board_size=" << c.env_board_size << "\nenv_go_komi=" << c.env_go_komi << "\nenv_go_ko_rule=" << c.env_go_ko_rule << "\n";
    return copyOut(o.str(), buf, cap);
}

// ---- network ----
long mzo_net_param_count(const NetDesc* d) { return static_cast<long>(Net::rawParamCount(*d)); }
void mzo_net_generate(const NetDesc* d, unsigned long long seed, float* out) { Net::generateRaw(*d, seed, out); }
void* mzo_net_create(const NetDesc* d, const float* raw, long n) { return Net::create(*d, raw, static_cast<size_t>(n)).release(); }
void mzo_net_destroy(void* net) { delete static_cast<Net*>(net); }
void mzo_net_forward_az(void* net, const float* feat, int B, float* policy, float* logit, float* value)
{
    static_cast<Net*>(net)->forwardAZ(feat, B, policy, logit, value);
}
void mzo_net_initial(void* net, const float* feat, int B, float* policy, float* logit, float* value, float* hidden)
{
    static_cast<Net*>(net)->initialMZ(feat, B, policy, logit, value, hidden);
}
void mzo_net_recurrent(void* net, const float* hidden_in, const float* action, int B, float* policy, float* logit, float* value, float* reward,
                       float* hidden_out)
{
    static_cast<Net*>(net)->recurrentMZ(hidden_in, action, B, policy, logit, value, reward, hidden_out);
}

// ---- environment ----
void* mzo_env_create(const char* conf)
{
    Config c;
    if (!c.loadFromString(conf)) { return nullptr; }
    if (c.env_board_size == 0) { c.setUpEnv(); }
    static Random env_rng; // only the Atari-shaped env draws from it
    return createEnv(c, &env_rng).release();
}
void mzo_env_destroy(void* e) { delete static_cast<Env*>(e); }
void mzo_env_reset(void* e) { static_cast<Env*>(e)->reset(); }
int mzo_env_act(void* e, int action_id, int player) { return static_cast<Env*>(e)->act(Action(action_id, static_cast<Player>(player))) ? 1 : 0; }
int mzo_env_turn(void* e) { return static_cast<Env*>(e)->getTurn(); }
int mzo_env_is_terminal(void* e) { return static_cast<Env*>(e)->isTerminal() ? 1 : 0; }
float mzo_env_eval_score(void* e, int is_resign) { return static_cast<Env*>(e)->getEvalScore(is_resign != 0); }
int mzo_env_policy_size(void* e) { return static_cast<Env*>(e)->getPolicySize(); }
int mzo_env_num_input_channels(void* e) { return static_cast<Env*>(e)->getNumInputChannels(); }
int mzo_env_board_size(void* e) { return static_cast<Env*>(e)->getBoardSize(); }
float mzo_env_reward(void* e) { return static_cast<Env*>(e)->getReward(); }
float mzo_invert_value(float v) { return invertValue(v); }
int mzo_conv_selftest(int cin, int cout, int H, int W, int stride, int with_skip, uint64_t seed) { return convSelfTest(cin, cout, H, W, stride, with_skip, seed); }
float mzo_transform_value(float v) { return transformValue(v); }
int mzo_env_seed(void* e)
{
    for (auto& t : static_cast<Env*>(e)->loaderTags()) { if (t.first == "SD") { return std::stoi(t.second); } }
    return 0;
}
void mzo_env_legal_mask(void* e, unsigned char* out)
{
    Env* env = static_cast<Env*>(e);
    for (int a = 0; a < env->getPolicySize(); ++a) { out[a] = env->isLegalAction(Action(a, env->getTurn())) ? 1 : 0; }
}
int mzo_env_features(void* e, int rotation, float* out)
{
    std::vector<float> f = static_cast<Env*>(e)->getFeatures(static_cast<Rotation>(rotation));
    memcpy(out, f.data(), f.size() * sizeof(float));
    return static_cast<int>(f.size());
}
int mzo_env_action_features(void* e, int action_id, int player, float* out)
{
    std::vector<float> f = static_cast<Env*>(e)->getActionFeatures(Action(action_id, static_cast<Player>(player)));
    memcpy(out, f.data(), f.size() * sizeof(float));
    return static_cast<int>(f.size());
}

// ---- bare search tree driven with injected candidates (kernel parity) ----
void* mzo_tree_create(const char* conf, long tree_node_size)
{
    Tree* t = new Tree();
    if (conf && *conf && !t->cfg.loadFromString(conf)) { delete t; return nullptr; }
    t->mcts = std::make_unique<MCTS>(&t->cfg, &t->rng, static_cast<uint64_t>(tree_node_size));
    t->mcts->reset();
    return t;
}
void mzo_tree_destroy(void* t) { delete static_cast<Tree*>(t); }
void mzo_tree_reset(void* tp, int root_player)
{
    Tree* t = static_cast<Tree*>(tp);
    t->mcts->reset();
    t->mcts->root()->action_ = Action(-1, static_cast<Player>(root_player));
}
// start < 0: select from root; else path = root + PUCT path below node index `start` (gumbel_zero.cpp:83-85)
int mzo_tree_select(void* tp, int start, int* path_out, int cap)
{
    Tree* t = static_cast<Tree*>(tp);
    if (start < 0) {
        t->path = t->mcts->select();
    } else {
        t->path = t->mcts->selectFromNode(&t->mcts->nodes_[start]);
        t->path.insert(t->path.begin(), t->mcts->root());
    }
    int n = static_cast<int>(t->path.size());
    for (int i = 0; i < n && i < cap; ++i) { path_out[i] = t->mcts->indexOf(t->path[i]); }
    return n;
}
void mzo_tree_expand_backup(void* tp, int k, const int* action_ids, int player, const float* policy, const float* logit, float value, float reward)
{
    Tree* t = static_cast<Tree*>(tp);
    if (k > 0) {
        std::vector<ActionCandidate> c;
        for (int i = 0; i < k; ++i) { c.emplace_back(Action(action_ids[i], static_cast<Player>(player)), policy[i], logit[i]); }
        t->mcts->expand(t->path.back(), c);
    }
    t->mcts->backup(t->path, value, reward);
}
void mzo_tree_set_child_policy(void* tp, int node, float policy, float logit, float noise)
{
    MCTSNode& n = static_cast<Tree*>(tp)->mcts->nodes_[node];
    n.policy_ = policy;
    n.policy_logit_ = logit;
    n.policy_noise_ = noise;
}
int mzo_tree_num_nodes(void* tp) { return static_cast<int>(static_cast<Tree*>(tp)->mcts->currentNodeSize()); }
void mzo_tree_dump(void* tp, int n, int* action, int* player, int* num_children, int* first_child, float* mean, float* count, float* policy,
                   float* logit, float* noise, float* value, float* reward)
{
    Tree* t = static_cast<Tree*>(tp);
    for (int i = 0; i < n; ++i) {
        const MCTSNode& nd = t->mcts->nodes_[i];
        action[i] = nd.action_.getActionID();
        player[i] = nd.action_.getPlayer();
        num_children[i] = nd.num_children_;
        first_child[i] = nd.first_child_;
        mean[i] = nd.mean_;
        count[i] = nd.count_;
        policy[i] = nd.policy_;
        logit[i] = nd.policy_logit_;
        noise[i] = nd.policy_noise_;
        value[i] = nd.value_;
        reward[i] = nd.reward_;
    }
}
int mzo_tree_value_bound(void* tp, float* lo, float* hi)
{
    auto& m = static_cast<Tree*>(tp)->mcts->tree_value_bound_;
    if (m.empty()) { return 0; }
    *lo = m.begin()->first;
    *hi = m.rbegin()->first;
    return static_cast<int>(m.size());
}

// compressString (utils.h:35-91) of n bytes: returns the hex length (the string itself when cap is large enough)
int mzo_compress_string(const char* data, int n, char* buf, int cap) { return copyOut(compressString(std::string(data, static_cast<size_t>(n))), buf, cap); }

// ---- learner-side sampler ----
void* mzo_loader_create(const char* conf)
{
    Config c;
    if (!c.loadFromString(conf)) { return nullptr; }
    return new DataLoaderOracle(c);
}
void mzo_loader_destroy(void* l) { delete static_cast<DataLoaderOracle*>(l); }
int mzo_loader_add(void* l, const char* line) { return static_cast<DataLoaderOracle*>(l)->addEnvString(line) ? 1 : 0; }
void mzo_loader_finish(void* l) { static_cast<DataLoaderOracle*>(l)->finishLoading(); }
void mzo_loader_load_file(void* l, const char* path) { static_cast<DataLoaderOracle*>(l)->loadDataFromFile(path); }
int mzo_loader_num_data(void* l) { return static_cast<DataLoaderOracle*>(l)->num_data_; }
int mzo_loader_num_games(void* l) { return static_cast<int>(static_cast<DataLoaderOracle*>(l)->env_loaders_.size()); }
void mzo_loader_sample(void* l, float* features, float* action_features, float* policy, float* value, float* reward, float* loss_scale, int* sampled_index)
{
    DataLoaderOracle::Batch b{features, action_features, policy, value, reward, loss_scale, sampled_index};
    static_cast<DataLoaderOracle*>(l)->sampleData(b);
}
void mzo_loader_update_priority(void* l, const int* sampled_index, const float* batch_values) { static_cast<DataLoaderOracle*>(l)->updatePriority(sampled_index, batch_values); }
// the record state machine alone (pinned to the reference's SGFLoader): renders tags and actions of `content` parsed with SGF move values
int mzo_sgf_parse(const char* content, char* buf, int cap)
{
    RecordLoader r;
    std::ostringstream o;
    const bool ok = r.loadFromString(content, -1, true);
    o << (ok ? "ok" : "fail") << "|";
    for (auto& t : r.tags_.items) { o << t.first << "=" << t.second << ";"; }
    o << "|";
    for (size_t i = 0; i < r.sgf_moves_.size(); ++i) {
        o << r.sgf_moves_[i].first << ":" << r.sgf_moves_[i].second << "{";
        for (auto& t : r.actions_[i].info.items) { o << t.first << "=" << t.second << ";"; }
        o << "}";
    }
    return copyOut(o.str(), buf, cap);
}
// VectorMap semantics: ops = lines "set k v" | "insert k v" | "erase k"; renders k[v]...
int mzo_tagmap_apply(const char* ops, char* buf, int cap)
{
    TagMap m;
    std::istringstream iss(ops);
    std::string line;
    while (std::getline(iss, line)) {
        std::istringstream ls(line);
        std::string op, k, v;
        ls >> op >> k >> v;
        if (op == "set") { m[k] = v; }
        else if (op == "insert") { m.insert(k, v); }
        else if (op == "erase") { m.erase(k); }
    }
    std::ostringstream o;
    for (auto& t : m.items) { o << t.first << "[" << t.second << "]"; }
    return copyOut(o.str(), buf, cap);
}
int mzo_sgf_coords(int action_id, int board_size, const char* coord, const char* sgf, int* out)
{
    out[0] = boardCoordinateStringToActionID(coord, board_size);
    out[1] = sgfStringToActionID(sgf, board_size);
    return 0;
}
int mzo_sgf_strings(int action_id, int board_size, char* buf, int cap)
{
    return copyOut(actionIDToBoardCoordinateString(action_id, board_size) + "|" + actionIDToSGFString(action_id, board_size), buf, cap);
}

// ---- self-play group ----
void* mzo_group_create(const char* conf, const NetDesc* d, const float* raw, long n)
{
    Config c;
    if (!c.loadFromString(conf)) { return nullptr; }
    if (c.env_board_size == 0) { c.setUpEnv(); }
    Group* g = new Group(c, *d, raw, static_cast<size_t>(n));
    if (!g->net_) { delete g; return nullptr; }
    return g;
}
void mzo_group_destroy(void* g) { delete static_cast<Group*>(g); }
void mzo_group_set_trace(void* g, int on) { static_cast<Group*>(g)->trace_ = on != 0; }
// runs n cycles while the group is started (a `stop` command makes this a no-op like ActorGroup::run's `if (!running_) continue`, actor_group.cpp:140); cycles run
int mzo_group_cycles(void* g, int n)
{
    Group* grp = static_cast<Group*>(g);
    if (!grp->running_) { return 0; }
    for (int i = 0; i < n; ++i) { grp->cycle(); }
    return n;
}
// one protocol line between two cycles; raw/n: the parameters of the file a load_model line names
int mzo_group_command(void* g, const char* line, const float* raw, long n) { return static_cast<Group*>(g)->command(line, raw, static_cast<size_t>(n)); }
unsigned long long mzo_group_num_cycles(void* g) { return static_cast<Group*>(g)->cycles_; }
unsigned long long mzo_group_leaf_evals(void* g) { return static_cast<Group*>(g)->q_.leaf_evals; }
unsigned long long mzo_group_games(void* g) { return static_cast<Group*>(g)->games_; }
int mzo_group_num_lines(void* g) { return static_cast<int>(static_cast<Group*>(g)->lines_.size()); }
int mzo_group_line(void* g, int i, char* buf, int cap) { return copyOut(static_cast<Group*>(g)->lines_[i], buf, cap); }
// record of actor i's game as it stands (unfinished games included): lets a test compare move / visit distributions without playing to the end
int mzo_group_peek_record(void* g, int i, char* buf, int cap) { return copyOut(static_cast<Group*>(g)->actors_[i]->getRecord({}), buf, cap); }
int mzo_group_num_trace(void* g) { return static_cast<int>(static_cast<Group*>(g)->trace_lines_.size()); }
int mzo_group_trace(void* g, int i, char* buf, int cap) { return copyOut(static_cast<Group*>(g)->trace_lines_[i], buf, cap); }

} // extern "C"
