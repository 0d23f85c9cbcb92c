// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Config, RNG, rotation, player helpers.
#include "oracle.h"
#include <algorithm>
#include <cassert>
#include <cmath>
#include <functional>
#include <iostream>
#include <numeric>
#include <sstream>

namespace mzo {

// ---- config: ref config/configure_loader.h:12-17 (istream extraction, nothing left over),
//      configure_loader.cpp:9-24 (bool: TRUE/1/FALSE/0 case-insensitive; string verbatim),
//      configure_loader.cpp:51-62 (':'-separated), :92-117 (key=value, '#' comment, trim) ----
template <class T>
static bool setParam(T& ref, const std::string& value)
{
    std::istringstream iss(value);
    iss >> ref;
    return (iss && iss.rdbuf()->in_avail() == 0);
}
static bool setParamBool(bool& ref, const std::string& value)
{
    std::string tmp = value;
    std::transform(tmp.begin(), tmp.end(), tmp.begin(), ::toupper);
    if (tmp != "TRUE" && tmp != "1" && tmp != "FALSE" && tmp != "0") { return false; }
    ref = (tmp == "TRUE" || tmp == "1");
    return true;
}
static void trim(std::string& s)
{
    if (s.empty()) { return; }
    s.erase(0, s.find_first_not_of(" \t"));
    s.erase(s.find_last_not_of(" \t") + 1);
}

bool Config::loadFromString(const std::string& conf)
{
    if (conf.empty()) { return false; }
    std::map<std::string, std::function<bool(const std::string&)>> p;
#define P_(name) p[#name] = [this](const std::string& v) { return setParam(name, v); }
#define PB(name) p[#name] = [this](const std::string& v) { return setParamBool(name, v); }
#define PS(name) p[#name] = [this](const std::string& v) { name = v; return true; }
    P_(program_seed); PB(program_auto_seed); PB(program_quiet);
    P_(actor_num_simulation); P_(actor_mcts_puct_base); P_(actor_mcts_puct_init); P_(actor_mcts_reward_discount);
    PB(actor_mcts_value_rescale); P_(actor_mcts_value_flipping_player); P_(actor_mcts_think_batch_size); P_(actor_mcts_think_time_limit);
    PB(actor_select_action_by_count); PB(actor_select_action_by_softmax_count); P_(actor_select_action_softmax_temperature);
    PB(actor_select_action_softmax_temperature_decay); PB(actor_use_random_rotation_features); PB(actor_use_dirichlet_noise);
    P_(actor_dirichlet_noise_alpha); P_(actor_dirichlet_noise_epsilon); PB(actor_use_gumbel); PB(actor_use_gumbel_noise);
    P_(actor_gumbel_sample_size); P_(actor_gumbel_sigma_visit_c); P_(actor_gumbel_sigma_scale_c); P_(actor_resign_threshold);
    P_(zero_num_threads); P_(zero_num_parallel_games); P_(zero_disable_resign_ratio); P_(zero_actor_intermediate_sequence_length);
    PS(zero_actor_ignored_command); P_(learner_muzero_unrolling_step); P_(learner_n_step_return);
    P_(zero_num_games_per_iteration); P_(zero_replay_buffer); PB(learner_use_per); P_(learner_per_alpha); P_(learner_per_init_beta); P_(learner_batch_size);
    PS(nn_file_name); PS(nn_type_name); P_(env_board_size); P_(env_go_komi); PS(env_go_ko_rule);
    PS(env_game); PB(atari_init_q); P_(oracle_throughput_threads); PS(env_atari_name); P_(env_atari_episode_length);
#undef P_
#undef PB
#undef PS
    std::string line;
    std::istringstream iss(conf);
    while (std::getline(iss, line, ':')) {
        if (line.empty() || line[0] == '#') { continue; }
        std::string key = line.substr(0, line.find("="));
        std::string value = line.substr(line.find("=") + 1);
        if (value.find("#") != std::string::npos) { value = value.substr(0, value.find("#")); }
        trim(key);
        trim(value);
        if (!p.count(key)) { return false; }
        if (!p[key](value)) { return false; }
    }
    return true;
}

void Config::setUpEnv()
{
    // ref: tictactoe.h:36 (3), othello.h:48 (8), go.h:79-83 (9)
    if (env_game == "tictactoe") { env_board_size = 3; }
    else if (env_game == "othello") { env_board_size = 8; }
    else if (env_game == "go") { env_board_size = 9; }
}

// ---- RNG: ref utils/random.h:15-36 ----
std::vector<float> Random::randDirichlet(float alpha, int size)
{
    std::vector<float> dirichlet;
    std::gamma_distribution<float> gamma_distribution(alpha);
    for (int i = 0; i < size; ++i) { dirichlet.emplace_back(gamma_distribution(generator_)); }
    float sum = std::accumulate(dirichlet.begin(), dirichlet.end(), 0.0f);
    if (sum < std::numeric_limits<float>::min()) { return dirichlet; }
    for (int i = 0; i < size; ++i) { dirichlet[i] /= sum; }
    return dirichlet;
}

std::vector<float> Random::randGumbel(int size)
{
    std::extreme_value_distribution<float> gumbel_distribution(0.0, 1.0);
    std::vector<float> gumbel;
    for (int i = 0; i < size; ++i) {
        float value = gumbel_distribution(generator_);
        while (std::isinf(value)) { value = gumbel_distribution(generator_); }
        gumbel.emplace_back(value);
    }
    return gumbel;
}

// ---- rotation: ref utils/rotation.h:21-29 (reversed table), :51-93 (float centre arithmetic, truncation) ----
const Rotation reversed_rotation[kRotateSize] = {kRotationNone, kRotation270, kRotation180, kRotation90, kHorizontalRotation,
                                                 kHorizontalRotation90, kHorizontalRotation180, kHorizontalRotation270};

int getPositionByRotating(Rotation rotation, int original_pos, int board_size)
{
    if (original_pos == board_size * board_size) { return original_pos; }
    const float center = (board_size - 1) / 2.0;
    float x = original_pos % board_size - center;
    float y = original_pos / board_size - center;
    float rotation_x = x, rotation_y = y;
    switch (rotation) {
        case kRotationNone: rotation_x = x, rotation_y = y; break;
        case kRotation90: rotation_x = y, rotation_y = -x; break;
        case kRotation180: rotation_x = -x, rotation_y = -y; break;
        case kRotation270: rotation_x = -y, rotation_y = x; break;
        case kHorizontalRotation: rotation_x = x, rotation_y = -y; break;
        case kHorizontalRotation90: rotation_x = -y, rotation_y = -x; break;
        case kHorizontalRotation180: rotation_x = -x, rotation_y = y; break;
        case kHorizontalRotation270: rotation_x = y, rotation_y = x; break;
        default: assert(false); break;
    }
    int new_pos = (rotation_y + center) * board_size + (rotation_x + center);
    return new_pos;
}

// ---- players: ref environment/base/base_env.cpp:5-42 ----
char playerToChar(Player p)
{
    switch (p) {
        case kPlayerNone: return 'N';
        case kPlayer1: return 'B';
        case kPlayer2: return 'W';
        default: return '?';
    }
}
Player charToPlayer(char c)
{
    switch (c) {
        case 'N': return kPlayerNone;
        case 'B':
        case 'b': return kPlayer1;
        case 'W':
        case 'w': return kPlayer2;
        default: return kPlayerSize;
    }
}
Player getNextPlayer(Player player, int num_player)
{
    if (num_player == 1) { return player; }
    else if (num_player == 2) { return (player == kPlayer1 ? kPlayer2 : kPlayer1); }
    return kPlayerNone;
}
Player getPreviousPlayer(Player player, int num_player)
{
    if (num_player <= 2) { return getNextPlayer(player, num_player); }
    return kPlayerNone;
}

std::vector<Action> Env::getLegalActions() const
{
    std::vector<Action> actions;
    for (int pos = 0; pos < getPolicySize(); ++pos) {
        Action action(pos, turn_);
        if (!isLegalAction(action)) { continue; }
        actions.push_back(action);
    }
    return actions;
}

} // namespace mzo
