// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the rlglab/minizero self-play hot path (the reference's
// own algorithm, restated plainly; not the product).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
// product (minizero_amd/) never includes, links or calls anything in oracle/.
//
// Pinning status (see DESIGN.md "Oracle"):
//   * network math           — pinned: tests/golden/nn_*.npz are outputs of the
//                              reference's own Python modules (minizero/network/py)
//                              imported in the build container.
//   * RNG / rotation / config— pinned: oracle/_ref builds the reference's own
//                              utils/random.{h,cpp}, utils/rotation.h and
//                              config/*.cpp in place and the restatement is
//                              compared against it.
//   * search / actor / envs  — PARITY UNPINNED: minizero/actor/*.cpp and
//                              minizero/environment/** include Boost headers
//                              (utils/utils.h:4-6, utils/time_system.h:3) that this
//                              image does not have, so they cannot be compiled
//                              without writing stand-ins; the reference ships no
//                              golden vectors for them.  The
//                              restatement follows the cited lines one by one.
//
// All `ref:` citations are relative to /root/reference/minizero/.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <random>
#include <deque>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace mzo {

// ----------------------------------------------------------------------------
// config  (ref: config/configuration.cpp:7-90, config/configure_loader.cpp)
// ----------------------------------------------------------------------------
struct Config {
    int program_seed = 0;
    bool program_auto_seed = false;
    bool program_quiet = false;
    int actor_num_simulation = 50;
    float actor_mcts_puct_base = 19652;
    float actor_mcts_puct_init = 1.25;
    float actor_mcts_reward_discount = 1.0f;
    int actor_mcts_think_batch_size = 1;
    float actor_mcts_think_time_limit = 0;
    bool actor_mcts_value_rescale = false;
    char actor_mcts_value_flipping_player = 'W';
    bool actor_select_action_by_count = false;
    bool actor_select_action_by_softmax_count = true;
    float actor_select_action_softmax_temperature = 1.0f;
    bool actor_select_action_softmax_temperature_decay = false;
    bool actor_use_random_rotation_features = true;
    bool actor_use_dirichlet_noise = true;
    float actor_dirichlet_noise_alpha = 0.03f;
    float actor_dirichlet_noise_epsilon = 0.25f;
    bool actor_use_gumbel = false;
    bool actor_use_gumbel_noise = false;
    int actor_gumbel_sample_size = 16;
    float actor_gumbel_sigma_visit_c = 50;
    float actor_gumbel_sigma_scale_c = 1;
    float actor_resign_threshold = -0.9f;
    int zero_num_threads = 4;
    int zero_num_parallel_games = 32;
    float zero_disable_resign_ratio = 0.1;
    int zero_actor_intermediate_sequence_length = 0;
    std::string zero_actor_ignored_command = "reset_actors";
    int learner_muzero_unrolling_step = 5;
    int learner_n_step_return = 0;
    // learner-side sampler (ref configuration.cpp:40-67): replay buffer size and prioritised replay
    int zero_num_games_per_iteration = 2000;
    int zero_replay_buffer = 20;
    bool learner_use_per = false;
    float learner_per_alpha = 1.0f;
    float learner_per_init_beta = 1.0f;
    int learner_batch_size = 1024;
    std::string nn_file_name = "";
    std::string nn_type_name = "alphazero";
    int env_board_size = 0;
    float env_go_komi = 7.5;
    std::string env_go_ko_rule = "positional";
    // not a reference key: the reference picks the game at compile time (-D<GAME>,
    // environment/environment.h:5-110) and the ATARI init-Q rule with it (actor/mcts.cpp:211-216).
    std::string env_game = "tictactoe";
    bool atari_init_q = false;
    int oracle_throughput_threads = 0; // oracle-only: T > 1 = the reference with zero_num_threads = T slave threads, actors statically partitioned (o_actor.cpp Group)
    std::string env_atari_name = "ms_pacman";
    int env_atari_episode_length = 1000; // synthetic Atari-shaped env: steps per episode

    // "k=v:k=v" string, later keys win (ref: configure_loader.cpp:51-117). Returns false on bad key/value.
    bool loadFromString(const std::string& s);
    void setUpEnv(); // ref: <game>.h setUpEnv(): default board size per game
};

// ----------------------------------------------------------------------------
// RNG  (ref: utils/random.h:9-41) — one generator per host thread in the reference
// ----------------------------------------------------------------------------
struct Random {
    std::mt19937 generator_;
    std::uniform_int_distribution<int> int_distribution_;
    std::uniform_real_distribution<double> real_distribution_;
    uint64_t draws_ = 0; // bookkeeping for traces only
    void seed(int s) { generator_.seed(s); }
    int randInt() { return int_distribution_(generator_); }
    double randReal(double range = 1.0f) { return real_distribution_(generator_) * range; }
    std::vector<float> randDirichlet(float alpha, int size);
    std::vector<float> randGumbel(int size);
};

// rotation (ref: utils/rotation.h:9-93)
enum Rotation { kRotationNone, kRotation90, kRotation180, kRotation270, kHorizontalRotation, kHorizontalRotation90,
                kHorizontalRotation180, kHorizontalRotation270, kRotateSize };
extern const Rotation reversed_rotation[kRotateSize];
int getPositionByRotating(Rotation rotation, int original_pos, int board_size);

// ----------------------------------------------------------------------------
// environment (ref: environment/base/base_env.h:74-114)
// ----------------------------------------------------------------------------
enum Player { kPlayerNone = 0, kPlayer1 = 1, kPlayer2 = 2, kPlayerSize = 3 };
char playerToChar(Player p);
Player charToPlayer(char c);
Player getNextPlayer(Player player, int num_player);
Player getPreviousPlayer(Player player, int num_player);

struct Action {
    int action_id_ = -1;
    Player player_ = kPlayerNone;
    Action() {}
    Action(int id, Player p) : action_id_(id), player_(p) {}
    int getActionID() const { return action_id_; }
    Player getPlayer() const { return player_; }
};

class Env {
public:
    virtual ~Env() = default;
    virtual std::unique_ptr<Env> clone() const = 0;
    virtual void reset() = 0;
    virtual bool act(const Action& action) = 0;
    virtual bool isLegalAction(const Action& action) const = 0;
    virtual bool isTerminal() const = 0;
    virtual float getReward() const { return 0.0f; }
    virtual float getEvalScore(bool is_resign = false) const = 0;
    virtual std::vector<float> getFeatures(Rotation rotation = kRotationNone) const = 0;
    virtual std::vector<float> getActionFeatures(const Action& action, Rotation rotation = kRotationNone) const = 0;
    virtual int getNumInputChannels() const = 0;
    virtual int getBoardSize() const = 0;
    virtual int getPolicySize() const = 0;
    virtual std::string name() const = 0;
    virtual int getNumPlayer() const { return 2; }
    virtual std::vector<std::pair<std::string, std::string>> loaderTags() const = 0; // SZ / KM after OBS
    // ref base_env.h:105-106: observation strings of the game so far (only Atari keeps any); atari.h:84: lives before action i (index i)
    virtual const std::vector<std::string>& getObservationHistory() const { static const std::vector<std::string> none; return none; }
    virtual std::vector<int> getLivesHistory() const { return {}; }
    virtual void resetWithSeed(int) { reset(); } // atari.h:53: reset(seed)
    int getRotateAction(int action_id, Rotation rotation) const { return getPositionByRotating(rotation, action_id, getBoardSize()); }
    Player getTurn() const { return turn_; }
    const std::vector<Action>& getActionHistory() const { return actions_; }
    Action nextPlayerAction(const Action& a) const; // unused helper
    std::vector<Action> getLegalActions() const;

protected:
    Player turn_ = kPlayer1;
    std::vector<Action> actions_;

public:
    Random* rng_ = nullptr; // only the Atari-shaped env draws from it (ref atari.h:54: reset(Random::randInt()))
};
std::unique_ptr<Env> createEnv(const Config& cfg, Random* rng = nullptr);
// ref utils/utils.h:35-91: gzip member (boost::iostreams::gzip_compressor defaults) of `s`, then two lower-case hex digits per byte; "" -> ""
std::string compressString(const std::string& s);

// ----------------------------------------------------------------------------
// learner-side sampler (o_loader.cpp): record loaders + replay buffer + DataLoader, one slave thread
// ----------------------------------------------------------------------------
struct TagMap { // utils/vector_map.h
    std::vector<std::pair<std::string, std::string>> items;
    std::string& operator[](const std::string& key);
    const std::string& get(const std::string& key) const;
    bool count(const std::string& key) const;
    bool insert(const std::string& key, const std::string& value);
    void erase(const std::string& key);
};
int sgfStringToActionID(const std::string& sgf_string, int board_size);
std::string actionIDToSGFString(int action_id, int board_size);
int boardCoordinateStringToActionID(const std::string& s, int board_size);
std::string actionIDToBoardCoordinateString(int action_id, int board_size);

class RecordLoader { // the record state machine shared by utils/sgf_loader.cpp and base_env.h
public:
    struct Pair { Action action; TagMap info; };
    // sgf_moves: move values are SGF coordinates kept as strings (SGFLoader); else "digits = action id" (BaseEnvLoader)
    bool loadFromString(const std::string& content, int default_board_size, bool sgf_moves);
    TagMap tags_;
    std::vector<Pair> actions_;
    std::vector<std::pair<std::string, std::string>> sgf_moves_; // SGFLoader::SGFAction (player key, board coordinate)
    int board_size_ = 0;
};

class GameLoader : public RecordLoader { // EnvironmentLoader of the configured game
public:
    explicit GameLoader(const Config* cfg);
    bool load(const std::string& content);
    std::pair<int, int> getDataRange() const;
    std::vector<float> getFeatures(int pos, Rotation rotation, Random* rng) const;
    std::vector<float> getActionFeatures(int pos, Rotation rotation, Random* rng) const;
    std::vector<float> getPolicy(int pos, Rotation rotation) const;
    std::vector<float> getValue(int pos) const;
    std::vector<float> getReward(int pos) const;
    float getPriority(int pos) const;
    bool setActionPairInfo(int pos, const std::string& tag, const std::string& value);
    int policySize() const;

private:
    std::unique_ptr<Env> newEnv() const;
    int rotateAction(int action_id, Rotation rotation) const;
    std::vector<float> getFeaturesByReplay(int pos) const;
    void addObservations(const std::string& compressed_obs);
    float baseValue(int pos) const;
    float baseReward(int pos) const;
    float calculateNStepValue(int pos) const;
    const Config* cfg_;
    std::vector<std::string> observations_;
};

class DataLoaderOracle {
public:
    struct Batch { float *features, *action_features, *policy, *value, *reward, *loss_scale; int* sampled_index; };
    explicit DataLoaderOracle(const Config& cfg);
    bool addEnvString(const std::string& env_string);
    void finishLoading();
    void loadDataFromFile(const std::string& file_name);
    void sampleData(Batch& b);
    void updatePriority(const int* sampled_index, const float* batch_values);
    int num_data_ = 0;
    float game_priority_sum_ = 0.0f;
    std::deque<float> game_priorities_;
    std::deque<std::deque<float>> position_priorities_;
    std::deque<GameLoader> env_loaders_;
    Config cfg_;

private:
    void addData(const GameLoader& env_loader);
    int sampleIndex(const std::deque<float>& weight);
    float getLossScale(const std::pair<int, int>& p);
    void sampleOne(int batch_index, Batch& b);
    Random rng_;
};

// ----------------------------------------------------------------------------
// network (the math of network/py/*.py with BN folded; see o_nn.cpp)
// ----------------------------------------------------------------------------
struct NetDesc {
    // ref: network/py/create_network.py:6-18 argument order
    char game_name[64];
    int num_input_channels, input_channel_height, input_channel_width;
    int num_hidden_channels, hidden_channel_height, hidden_channel_width;
    int num_action_feature_channels, num_blocks, action_size, num_value_hidden_channels, discrete_value_size;
    int type; // 0 alphazero, 1 muzero, 2 muzero_atari
};
struct NetOutput { // ref: network/alphazero_network.h:13-26, muzero_network.h:14-31
    float value_ = 0, reward_ = 0;
    std::vector<float> policy_, policy_logits_, hidden_state_;
};
class Net {
public:
    virtual ~Net() = default;
    NetDesc desc;
    // raw (un-folded) parameters in state_dict order; see o_nn.cpp for the manifest
    static size_t rawParamCount(const NetDesc& d);
    static void generateRaw(const NetDesc& d, uint64_t seed, float* out); // deterministic synthetic weights
    static std::unique_ptr<Net> create(const NetDesc& d, const float* raw, size_t n);
    virtual void forwardAZ(const float* features, int batch, float* policy, float* logit, float* value) const = 0;
    virtual void initialMZ(const float* features, int batch, float* policy, float* logit, float* value, float* hidden) const = 0;
    virtual void recurrentMZ(const float* hidden_in, const float* action_plane, int batch, float* policy, float* logit, float* value,
                             float* reward, float* hidden_out) const = 0;
};
float mz_expf(float x);  // deterministic expf shared (by specification) with the HIP kernels
float mz_tanhf(float x);
float invertValue(float value); // ref utils/utils.h:102-108
int convSelfTest(int cin, int cout, int H, int Wd, int stride, int with_skip, uint64_t seed); // o_nn.cpp: the register-blocked convolution against the scalar chain (0 = every bit equal)
float transformValue(float value); // ref utils/utils.h:93-100

// ----------------------------------------------------------------------------
// search (ref: actor/tree.h, actor/mcts.{h,cpp}, actor/gumbel_zero.{h,cpp})
// ----------------------------------------------------------------------------
struct MCTSNode {
    Action action_;
    int num_children_ = 0;
    int first_child_ = -1; // index into the arena (the reference keeps a pointer)
    int hidden_state_data_index_ = -1;
    float mean_ = 0, count_ = 0, virtual_loss_ = 0, policy_ = 0, policy_logit_ = 0, policy_noise_ = 0, value_ = 0, reward_ = 0;
    void reset();
    void add(float value, float weight = 1.0f);
    bool isLeaf() const { return num_children_ == 0; }
    float getCountWithVirtualLoss() const { return count_ + virtual_loss_; }
};
struct ActionCandidate {
    Action action_;
    float policy_, policy_logit_;
    ActionCandidate(const Action& a, float p, float l) : action_(a), policy_(p), policy_logit_(l) {}
};
class MCTS {
public:
    MCTS(const Config* cfg, Random* rng, uint64_t tree_node_size) : cfg_(cfg), rng_(rng), tree_node_size_(tree_node_size) {}
    void reset();
    MCTSNode* root() { return &nodes_[0]; }
    const MCTSNode* root() const { return &nodes_[0]; }
    MCTSNode* child(const MCTSNode* n, int i) { return &nodes_[n->first_child_ + i]; }
    const MCTSNode* child(const MCTSNode* n, int i) const { return &nodes_[n->first_child_ + i]; }
    int indexOf(const MCTSNode* n) const { return int(n - &nodes_[0]); }
    float getNormalizedMean(const MCTSNode* n) const;
    float getNormalizedPUCTScore(const MCTSNode* n, int total_simulation, float init_q_value) const;
    bool isResign(const MCTSNode* selected) const;
    MCTSNode* selectChildByMaxCount(const MCTSNode* node);
    MCTSNode* selectChildBySoftmaxCount(const MCTSNode* node, float temperature = 1.0f, float value_threshold = 0.1f);
    std::string getSearchDistributionString() const;
    std::vector<MCTSNode*> select() { return selectFromNode(root()); }
    std::vector<MCTSNode*> selectFromNode(MCTSNode* start);
    void expand(MCTSNode* leaf, const std::vector<ActionCandidate>& cands);
    void backup(const std::vector<MCTSNode*>& path, float value, float reward = 0.0f);
    int getNumSimulation() const { return int(root()->count_); }
    bool reachMaximumSimulation() const { return getNumSimulation() == cfg_->actor_num_simulation + 1; }
    MCTSNode* selectChildByPUCTScore(const MCTSNode* node);
    float calculateInitQValue(const MCTSNode* node) const;
    void updateTreeValueBound(float old_value, float new_value);
    int storeHidden(const std::vector<float>& h) { hidden_.push_back(h); return int(hidden_.size()) - 1; }
    const std::vector<float>& hidden(int i) const { return hidden_[i]; }
    uint64_t currentNodeSize() const { return current_node_size_; }

    const Config* cfg_;
    Random* rng_;
    uint64_t tree_node_size_, current_node_size_ = 1;
    std::vector<MCTSNode> nodes_;
    std::map<float, int> tree_value_bound_;
    std::vector<std::vector<float>> hidden_;
};
class GumbelZero {
public:
    std::string getMCTSPolicy(const Config& cfg, MCTS& mcts) const;
    MCTSNode* decideActionNode(const Config& cfg, MCTS& mcts);
    std::vector<MCTSNode*> selection(MCTS& mcts);
    void sequentialHalving(const Config& cfg, MCTS& mcts);
    void sortCandidatesByScore(const Config& cfg, MCTS& mcts);
    int sample_size_ = 0, simulation_budget_ = 0;
    std::vector<MCTSNode*> candidates_;
};

// ----------------------------------------------------------------------------
// actor + group (ref: actor/{base_actor,zero_actor,actor_group}.cpp)
// ----------------------------------------------------------------------------
struct NetQueue; // per-cycle batch queue, see o_actor.cpp
class ZeroActor {
public:
    ZeroActor(const Config* cfg, Random* rng, const NetDesc* nd, NetQueue* q, uint64_t tree_node_size);
    void reset();
    void resetSearch();
    bool act(const Action& a);
    void beforeNNEvaluation();
    void afterNNEvaluation(const NetOutput& out);
    bool isSearchDone() const { return mcts_.reachMaximumSimulation(); }
    Action getSearchAction() const { return selected_node_->action_; }
    bool isResign() const { return enable_resign_ && mcts_.isResign(selected_node_); }
    std::string getRecord(const std::vector<std::pair<std::string, std::string>>& tags) const;
    bool isEnvTerminal() const { return env_->isTerminal(); }

    const Config* cfg_;
    Random* rng_;
    const NetDesc* nd_;
    NetQueue* q_;
    std::unique_ptr<Env> env_;
    MCTS mcts_;
    GumbelZero gumbel_zero_;
    bool enable_resign_ = true;
    int nn_evaluation_batch_id_ = -1;
    int slot_override_ = -1; // throughput mode: pre-assigned batch slot (threads push concurrently)
    Rotation feature_rotation_ = kRotationNone;
    MCTSNode* selected_node_ = nullptr;
    std::vector<MCTSNode*> node_path_;
    std::vector<std::vector<std::pair<std::string, std::string>>> action_info_history_;
    // trace hooks (test infrastructure): last path as arena indices, last candidate order
    std::vector<int> last_path_idx_;
    std::vector<int> last_cand_actions_;

private:
    std::unique_ptr<Env> getEnvironmentTransition(const std::vector<MCTSNode*>& path) const;
    std::vector<ActionCandidate> calculateAlphaZeroActionPolicy(const Env& env_transition, const NetOutput& out, Rotation rot);
    std::vector<ActionCandidate> calculateMuZeroActionPolicy(MCTSNode* leaf, const NetOutput& out);
    void addNoiseToNodeChildren(MCTSNode* node);
    MCTSNode* decideActionNode();
    std::vector<MCTSNode*> selection();
    std::vector<std::pair<std::string, std::string>> getActionInfo() const;
};

} // namespace mzo
