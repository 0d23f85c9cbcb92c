// ORACLE (test infrastructure): the learner-side sampler of the reference — ref learner/data_loader.{h,cpp} (ReplayBuffer, DataLoaderThread,
// DataLoader) and the record loaders it samples from: environment/base/base_env.h:116-312 (BaseEnvLoader: the record parser, getPolicy,
// getDataRange, getValue / getReward), go.h:120-140 + go.cpp:725-737, othello.h:64-84 + othello.cpp:264-276, tictactoe.h:38-52 +
// tictactoe.cpp:148-155 (board-game loaders), atari.h:106-131 + atari.cpp:171-292 (AtariEnvLoader: observations, n-step value, 601-bin targets,
// priorities).  One slave thread = the deterministic contract (thread id 0 seeds program_seed + 0, data_loader.cpp:106-110).
// PARITY: the record state machine and the tag map are pinned to the reference's own utils/sgf_loader.cpp / utils/vector_map.h compiled in
// place (oracle/_ref, tests/golden/ref_sgf_vectormap.json); everything else here is parity-unpinned restatement (environment/** needs Boost).
#include "oracle.h"
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <fstream>
#include <numeric>
#include <sstream>

namespace mzo {

// ---- utils/vector_map.h:11-106: insertion-ordered map; operator[] finds or appends ----
std::string& TagMap::operator[](const std::string& key)
{
    for (auto& kv : items) { if (kv.first == key) { return kv.second; } }
    items.emplace_back(key, std::string());
    return items.back().second;
}
const std::string& TagMap::get(const std::string& key) const
{
    static const std::string npos;
    for (auto& kv : items) { if (kv.first == key) { return kv.second; } }
    return npos;
}
bool TagMap::count(const std::string& key) const
{
    for (auto& kv : items) { if (kv.first == key) { return true; } }
    return false;
}
bool TagMap::insert(const std::string& key, const std::string& value) // vector_map.h:71-80: an existing key keeps its value
{
    if (count(key)) { return false; }
    items.emplace_back(key, value);
    return true;
}
void TagMap::erase(const std::string& key)
{
    for (size_t i = 0; i < items.size(); ++i) { if (items[i].first == key) { items.erase(items.begin() + i); return; } }
}

// ---- utils/sgf_loader.cpp:106-142 ----
int sgfStringToActionID(const std::string& sgf_string, int board_size)
{
    if (sgf_string.size() != 2) { return board_size * board_size; }
    int x = std::toupper(sgf_string[0]) - 'A';
    int y = (board_size - 1) - (std::toupper(sgf_string[1]) - 'A');
    return y * board_size + x;
}
std::string actionIDToSGFString(int action_id, int board_size)
{
    if (action_id == board_size * board_size) { return ""; }
    int x = action_id % board_size;
    int y = action_id / board_size;
    std::ostringstream oss;
    oss << static_cast<char>(x + 'a') << static_cast<char>(((board_size - 1) - y) + 'a');
    return oss.str();
}
int boardCoordinateStringToActionID(const std::string& s, int board_size)
{
    std::string tmp = s;
    std::transform(tmp.begin(), tmp.end(), tmp.begin(), ::toupper);
    if (tmp == "PASS") { return board_size * board_size; }
    if (s.size() < 2) { return -1; }
    int x = std::toupper(s[0]) - 'A' + (std::toupper(s[0]) > 'I' ? -1 : 0);
    int y = atoi(s.substr(1).c_str()) - 1;
    return y * board_size + x;
}
std::string actionIDToBoardCoordinateString(int action_id, int board_size)
{
    if (action_id == board_size * board_size) { return "PASS"; }
    int x = action_id % board_size;
    int y = action_id / board_size;
    std::ostringstream oss;
    oss << static_cast<char>(x + 'A' + (x >= 8)) << y + 1;
    return oss.str();
}

// ---- the record state machine: base_env.h:150-205 (== sgf_loader.cpp:26-83 up to how a move value becomes an action) ----
bool RecordLoader::loadFromString(const std::string& content, int default_board_size, bool sgf_moves)
{
    tags_ = TagMap();
    actions_.clear();
    sgf_moves_.clear();
    std::string key, value;
    int state = '(';
    bool accept_move = false;
    bool escape_next = false;
    int board_size = default_board_size;
    for (char c : content) {
        switch (state) {
            case '(': // wait until record start
                if (!accept_move) {
                    accept_move = (c == '(');
                } else {
                    state = (c == ';') ? c : 'x';
                    accept_move = false;
                }
                break;
            case ';': // store key
                if (c == ';') {
                    accept_move = true;
                } else if (c == '[' || c == ')') {
                    state = c;
                } else if (std::isgraph(c)) {
                    key += c;
                }
                break;
            case '[': // store value
                if (c == '\\' && !escape_next) {
                    escape_next = true;
                } else if (c != ']' || escape_next) {
                    value += c;
                    escape_next = false;
                } else { // ready to store key-value pair
                    if (accept_move) {
                        if (sgf_moves) { // SGFLoader (sgf_loader.cpp:62-64)
                            if (board_size == -1) { return false; }
                            actions_.emplace_back();
                            sgf_moves_.push_back({key, actionIDToBoardCoordinateString(sgfStringToActionID(value, board_size), board_size)});
                        } else { // BaseEnvLoader (base_env.h:182-185)
                            int action_id = value.size() && std::isdigit(value[0]) ? std::stoi(value) : sgfStringToActionID(value, board_size);
                            actions_.emplace_back();
                            actions_.back().action = Action(action_id, charToPlayer(key[0]));
                        }
                        accept_move = false;
                    } else if (actions_.size()) {
                        actions_.back().info[key] = std::move(value);
                    } else {
                        if (key == "SZ") { board_size = std::stoi(value); }
                        tags_[key] = std::move(value);
                    }
                    key.clear();
                    value.clear();
                    state = ';';
                }
                break;
            case ')': // end of record, do nothing
                break;
        }
    }
    board_size_ = board_size;
    return state == ')';
}

// ---- utils/utils.h:66-91 decompressString: hex -> gzip member -> bytes ----
static std::string decompressString(const std::string& hex)
{
    if (hex.empty()) { return hex; }
    std::string bin;
    for (size_t i = 0; i + 1 < hex.size(); i += 2) { bin += static_cast<char>(std::stoi(hex.substr(i, 2), nullptr, 16)); }
    z_stream zs{};
    if (inflateInit2(&zs, 15 + 16) != Z_OK) { return ""; } // gzip framing
    zs.next_in = reinterpret_cast<Bytef*>(&bin[0]);
    zs.avail_in = static_cast<uInt>(bin.size());
    std::string out;
    char buf[1 << 15];
    int rc = Z_OK;
    while (rc == Z_OK) {
        zs.next_out = reinterpret_cast<Bytef*>(buf);
        zs.avail_out = sizeof(buf);
        rc = inflate(&zs, Z_NO_FLUSH);
        out.append(buf, sizeof(buf) - zs.avail_out);
    }
    inflateEnd(&zs);
    return out;
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

// ---- the game loaders ----
GameLoader::GameLoader(const Config* cfg) : cfg_(cfg) {}

bool GameLoader::load(const std::string& content)
{
    Config c = *cfg_;
    if (c.env_board_size == 0) { c.setUpEnv(); }
    if (!RecordLoader::loadFromString(content, c.env_board_size, false)) { return false; }
    if (cfg_->env_game == "atari") { addObservations(tags_.get("OBS")); } // atari.cpp:179-184
    return true;
}

std::unique_ptr<Env> GameLoader::newEnv() const
{
    Config c = *cfg_;
    if (c.env_board_size == 0) { c.setUpEnv(); }
    if (cfg_->env_game != "atari" && board_size_ > 0) { c.env_board_size = board_size_; }
    return createEnv(c, nullptr);
}

int GameLoader::policySize() const
{
    const int n = board_size_;
    if (cfg_->env_game == "atari") { return 18; }
    if (cfg_->env_game == "tictactoe") { return 9; }
    return n * n + 1;
}

int GameLoader::rotateAction(int action_id, Rotation rotation) const
{
    if (cfg_->env_game == "atari") { return action_id; } // atari.h:122
    return getPositionByRotating(rotation, action_id, board_size_);
}

std::pair<int, int> GameLoader::getDataRange() const // base_env.h:267-276
{
    const std::string dlen = tags_.get("DLEN");
    if (dlen.empty()) { return {0, std::max(0, static_cast<int>(actions_.size()) - 1)}; }
    return {std::stoi(dlen), std::stoi(dlen.substr(dlen.find("-") + 1))};
}

std::vector<float> GameLoader::getFeatures(int pos, Rotation rotation, Random* rng) const
{
    if (cfg_->env_game != "atari") { // base_env.h:235-241: replay
        std::unique_ptr<Env> env = newEnv();
        for (int i = 0; i < std::min(pos, static_cast<int>(actions_.size())); ++i) { env->act(actions_[i].action); }
        return env->getFeatures(rotation);
    }
    // atari.cpp:199-221
    const int kRes = 96, kHist = 8, kActions = 18;
    std::vector<float> features;
    features.reserve(size_t(kHist) * 4 * kRes * kRes);
    int start = pos - kHist + 1, end = pos;
    for (int i = start; i <= end; ++i) {
        int action_id = (i - 1 < 0 ? 0 : (i - 1 >= static_cast<int>(actions_.size()) ? rng->randInt() % kActions : actions_[i - 1].action.getActionID()));
        std::vector<float> action_features(size_t(kRes) * kRes, action_id * 1.0f / kActions);
        features.insert(features.end(), action_features.begin(), action_features.end());
        if (i >= 0) {
            const std::string& observation = (i < static_cast<int>(observations_.size()) ? observations_[i] : observations_.back());
            if (observation.empty()) { return getFeaturesByReplay(pos); }
            for (const auto& o : observation) { features.push_back(static_cast<unsigned int>(static_cast<unsigned char>(o)) / 255.0f); }
        } else {
            std::vector<float> f(size_t(3) * kRes * kRes, 0.0f);
            features.insert(features.end(), f.begin(), f.end());
        }
    }
    return features;
}

std::vector<float> GameLoader::getFeaturesByReplay(int pos) const // atari.cpp:251-257
{
    std::unique_ptr<Env> env = newEnv();
    env->resetWithSeed(std::stoi(tags_.get("SD")));
    for (int i = 0; i < pos; ++i) { env->act(actions_[i].action); }
    return env->getFeatures(kRotationNone);
}

void GameLoader::addObservations(const std::string& compressed_obs) // atari.cpp:237-249
{
    observations_.assign(actions_.size() + 1, "");
    if (compressed_obs.empty()) { return; }
    const int obs_length = 3 * 96 * 96;
    std::string observations_str = decompressString(compressed_obs);
    int index = static_cast<int>(observations_.size());
    for (size_t end = observations_str.size(); end > 0 && index > 0; end -= obs_length) { observations_[--index] = observations_str.substr(end - obs_length, obs_length); }
}

std::vector<float> GameLoader::getActionFeatures(int pos, Rotation rotation, Random* rng) const
{
    const int size = static_cast<int>(actions_.size());
    if (cfg_->env_game == "atari") { // atari.cpp:223-235
        const int hidden_size = 36;
        std::vector<float> f(size_t(18) * hidden_size, 0.0f);
        const int action_id = pos < size ? actions_[pos].action.getActionID() : rng->randInt() % 18;
        std::fill(f.begin() + action_id * hidden_size, f.begin() + (action_id + 1) * hidden_size, 1.0f);
        return f;
    }
    const int n = board_size_;
    std::vector<float> f(size_t(n) * n, 0.0f);
    if (cfg_->env_game == "tictactoe") { // tictactoe.cpp:148-155
        const int action_id = (pos < size) ? rotateAction(actions_[pos].action.getActionID(), rotation) : rng->randInt() % static_cast<int>(f.size());
        f[action_id] = 1.0f;
        return f;
    }
    if (pos < size) { // go.cpp:725-737, othello.cpp:264-276
        const int a = actions_[pos].action.getActionID();
        if (a != n * n) { f[rotateAction(a, rotation)] = 1.0f; }
    } else {
        int action_id = rng->randInt() % (f.size() + 1);
        if (action_id < size && action_id < static_cast<int>(f.size())) { f[action_id] = 1.0f; } // (the reference's f[P] write lands outside its vector)
    }
    return f;
}

std::vector<float> GameLoader::getPolicy(int pos, Rotation rotation) const // base_env.h:243-265
{
    std::vector<float> policy(policySize(), 0.0f);
    if (pos < static_cast<int>(actions_.size())) {
        const std::string policy_distribution = actions_[pos].info.get("P");
        if (policy_distribution.empty()) {
            policy[rotateAction(actions_[pos].action.getActionID(), rotation)] = 1.0f;
        } else {
            std::string tmp;
            float total = 0.0f;
            std::istringstream iss(policy_distribution);
            while (std::getline(iss, tmp, ',')) {
                int position = rotateAction(std::stoi(tmp.substr(0, tmp.find(":"))), rotation);
                float count = std::stof(tmp.substr(tmp.find(":") + 1));
                policy[position] = count;
                total += count;
            }
            for (auto& p : policy) { p /= total; }
        }
    } else { // absorbing states
        std::fill(policy.begin(), policy.end(), 1.0f / policySize());
    }
    return policy;
}

float GameLoader::baseValue(int pos) const { return pos < static_cast<int>(actions_.size()) ? std::stof(actions_[pos].info.get("V")) : 0.0f; }   // base_env.h:278
float GameLoader::baseReward(int pos) const { return pos < static_cast<int>(actions_.size()) ? std::stof(actions_[pos].info.get("R")) : 0.0f; } // base_env.h:279

float GameLoader::calculateNStepValue(int pos) const // atari.cpp:259-277
{
    const int n_step = cfg_->learner_n_step_return;
    const float discount = cfg_->actor_mcts_reward_discount;
    size_t bootstrap_index = pos + n_step;
    float value = 0.0f;
    float n_step_value = ((bootstrap_index < actions_.size() && !actions_[bootstrap_index].info.count("L")) ? std::pow(discount, n_step) * baseValue(static_cast<int>(bootstrap_index)) : 0.0f);
    for (size_t index = pos; index < std::min(bootstrap_index, actions_.size()); ++index) {
        if (actions_[index].info.count("L") && std::stoi(actions_[index].info.get("L")) > 0) { return value; }
        float reward = baseReward(static_cast<int>(index));
        value += std::pow(discount, index - pos) * reward;
    }
    value += n_step_value;
    return value;
}

static std::vector<float> toDiscreteValue(float value) // atari.cpp:279-292
{
    const int kSize = 601;
    std::vector<float> discrete_value(kSize, 0.0f);
    int value_floor = floor(value);
    int value_ceil = ceil(value);
    int shift = kSize / 2;
    int value_floor_shift = std::min(std::max(value_floor + shift, 0), kSize - 1);
    int value_ceil_shift = std::min(std::max(value_ceil + shift, 0), kSize - 1);
    if (value_floor == value_ceil) {
        discrete_value[value_floor_shift] = 1.0f;
    } else {
        discrete_value[value_floor_shift] = value_ceil - value;
        discrete_value[value_ceil_shift] = value - value_floor;
    }
    return discrete_value;
}

std::vector<float> GameLoader::getValue(int pos) const
{
    if (cfg_->env_game == "atari") { return toDiscreteValue(pos < static_cast<int>(actions_.size()) ? transformValue(calculateNStepValue(pos)) : 0.0f); } // atari.h:115
    return {std::stof(tags_.get("RE"))}; // go.h:137, othello.h:79, tictactoe.h:47: getReturn()
}
std::vector<float> GameLoader::getReward(int pos) const
{
    if (cfg_->env_game == "atari") { return toDiscreteValue(pos < static_cast<int>(actions_.size()) ? transformValue(baseReward(pos)) : 0.0f); } // atari.h:116
    return {baseReward(pos)};
}
float GameLoader::getPriority(int pos) const
{
    if (cfg_->env_game == "atari") { return fabs(calculateNStepValue(pos) - baseValue(pos)) + 1e-6; } // atari.h:117
    return 1.0f;
}
bool GameLoader::setActionPairInfo(int pos, const std::string& tag, const std::string& value) // base_env.h:280-285
{
    if (pos >= static_cast<int>(actions_.size())) { return false; }
    actions_[pos].info[tag] = value;
    return true;
}

// ---- ReplayBuffer: data_loader.cpp:15-82 ----
void DataLoaderOracle::addData(const GameLoader& env_loader)
{
    std::pair<int, int> data_range = env_loader.getDataRange();
    std::deque<float> position_priorities(data_range.second + 1, 0.0f);
    float game_priority = 0.0f;
    for (int i = data_range.first; i <= data_range.second; ++i) {
        position_priorities[i] = std::pow((cfg_.learner_use_per ? env_loader.getPriority(i) : 1.0f), cfg_.learner_per_alpha);
        game_priority += position_priorities[i];
    }
    num_data_ += (data_range.second - data_range.first + 1);
    position_priorities_.push_back(position_priorities);
    game_priorities_.push_back(game_priority);
    env_loaders_.push_back(env_loader);
    const size_t replay_buffer_max_size = static_cast<size_t>(cfg_.zero_replay_buffer * cfg_.zero_num_games_per_iteration);
    while (position_priorities_.size() > replay_buffer_max_size) {
        data_range = env_loaders_.front().getDataRange();
        num_data_ -= (data_range.second - data_range.first + 1);
        position_priorities_.pop_front();
        game_priorities_.pop_front();
        env_loaders_.pop_front();
    }
}

int DataLoaderOracle::sampleIndex(const std::deque<float>& weight)
{
    std::discrete_distribution<> dis(weight.begin(), weight.end());
    return dis(rng_.generator_);
}

float DataLoaderOracle::getLossScale(const std::pair<int, int>& p)
{
    if (!cfg_.learner_use_per) { return 1.0f; }
    int env_id = p.first, pos = p.second;
    float prob = position_priorities_[env_id][pos] / game_priority_sum_;
    return std::pow((num_data_ * prob), (-cfg_.learner_per_init_beta));
}

DataLoaderOracle::DataLoaderOracle(const Config& cfg) : cfg_(cfg)
{
    if (cfg_.env_board_size == 0) { cfg_.setUpEnv(); }
    rng_.seed(cfg_.program_seed + 0); // DataLoaderThread::initialize, thread id 0
}

bool DataLoaderOracle::addEnvString(const std::string& env_string) // DataLoaderThread::addEnvironmentLoader
{
    GameLoader env_loader(&cfg_);
    if (!env_loader.load(env_string)) { return false; }
    addData(env_loader);
    return true;
}

void DataLoaderOracle::finishLoading() // tail of DataLoader::loadDataFromFile
{
    game_priority_sum_ = std::accumulate(game_priorities_.begin(), game_priorities_.end(), 0.0f);
}

void DataLoaderOracle::loadDataFromFile(const std::string& file_name)
{
    std::ifstream fin(file_name, std::ifstream::in);
    for (std::string content; std::getline(fin, content);) { addEnvString(content); }
    finishLoading();
}

void DataLoaderOracle::sampleOne(int batch_index, Batch& b) // setAlphaZeroTrainingData / setMuZeroTrainingData
{
    std::pair<int, int> p = {sampleIndex(game_priorities_), 0};
    p.second = sampleIndex(position_priorities_[p.first]);
    const int env_id = p.first, pos = p.second;
    const GameLoader& env_loader = env_loaders_[env_id];
    Rotation rotation = static_cast<Rotation>(rng_.randInt() % static_cast<int>(kRotateSize));
    float loss_scale = getLossScale(p);
    std::vector<float> features = env_loader.getFeatures(pos, rotation, &rng_);
    std::vector<float> action_features, policy, value, reward, tmp;
    if (cfg_.nn_type_name == "alphazero") {
        policy = env_loader.getPolicy(pos, rotation);
        value = env_loader.getValue(pos);
    } else {
        for (int step = 0; step <= cfg_.learner_muzero_unrolling_step; ++step) {
            if (step < cfg_.learner_muzero_unrolling_step) {
                tmp = env_loader.getActionFeatures(pos + step, rotation, &rng_);
                action_features.insert(action_features.end(), tmp.begin(), tmp.end());
            }
            tmp = env_loader.getPolicy(pos + step, rotation);
            policy.insert(policy.end(), tmp.begin(), tmp.end());
            tmp = env_loader.getValue(pos + step);
            value.insert(value.end(), tmp.begin(), tmp.end());
            if (step < cfg_.learner_muzero_unrolling_step) {
                tmp = env_loader.getReward(pos + step);
                reward.insert(reward.end(), tmp.begin(), tmp.end());
            }
        }
    }
    b.loss_scale[batch_index] = loss_scale;
    b.sampled_index[2 * batch_index] = p.first;
    b.sampled_index[2 * batch_index + 1] = p.second;
    std::copy(features.begin(), features.end(), b.features + features.size() * batch_index);
    if (b.action_features) { std::copy(action_features.begin(), action_features.end(), b.action_features + action_features.size() * batch_index); }
    std::copy(policy.begin(), policy.end(), b.policy + policy.size() * batch_index);
    std::copy(value.begin(), value.end(), b.value + value.size() * batch_index);
    if (b.reward) { std::copy(reward.begin(), reward.end(), b.reward + reward.size() * batch_index); }
}

void DataLoaderOracle::sampleData(Batch& b)
{
    for (int batch_index = 0; batch_index < cfg_.learner_batch_size; ++batch_index) { sampleOne(batch_index, b); }
}

void DataLoaderOracle::updatePriority(const int* sampled_index, const float* batch_values) // data_loader.cpp:233-253
{
    for (int batch_index = 0; batch_index < cfg_.learner_batch_size; ++batch_index) {
        int env_id = sampled_index[2 * batch_index];
        int pos_id = sampled_index[2 * batch_index + 1];
        GameLoader& env_loader = env_loaders_[env_id];
        for (int step = 0; step <= cfg_.learner_muzero_unrolling_step; ++step) {
            float new_value = invertValue(batch_values[step * cfg_.learner_batch_size + batch_index]);
            env_loader.setActionPairInfo(pos_id + step, "V", std::to_string(new_value));
        }
        position_priorities_[env_id][pos_id] = std::pow(env_loader.getPriority(pos_id), cfg_.learner_per_alpha);
    }
    for (size_t i = 0; i < game_priorities_.size(); ++i) { game_priorities_[i] = std::accumulate(position_priorities_[i].begin(), position_priorities_[i].end(), 0.0f); }
    game_priority_sum_ = std::accumulate(game_priorities_.begin(), game_priorities_.end(), 0.0f);
}

} // namespace mzo
