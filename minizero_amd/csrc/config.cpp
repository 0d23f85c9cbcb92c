// Table-driven parser for WorkerConfig (see config.h).
#include "config.h"
#include "common.h"
#include <algorithm>
#include <cstddef>
#include <sstream>

namespace mz {

namespace {
enum Type { T_INT, T_FLOAT, T_BOOL, T_CHAR, T_STRING };
struct Key { const char* name; Type type; size_t offset; };
#define K(name, type) {#name, type, offsetof(WorkerConfig, name)}
const Key kKeys[] = {
    K(program_seed, T_INT), K(program_auto_seed, T_BOOL), K(program_quiet, T_BOOL), K(actor_num_simulation, T_INT),
    K(actor_mcts_puct_base, T_FLOAT), K(actor_mcts_puct_init, T_FLOAT), K(actor_mcts_reward_discount, T_FLOAT),
    K(actor_mcts_think_batch_size, T_INT), K(actor_mcts_think_time_limit, T_FLOAT), K(actor_mcts_value_rescale, T_BOOL),
    K(actor_mcts_value_flipping_player, T_CHAR), K(actor_select_action_by_count, T_BOOL), K(actor_select_action_by_softmax_count, T_BOOL),
    K(actor_select_action_softmax_temperature, T_FLOAT), K(actor_select_action_softmax_temperature_decay, T_BOOL),
    K(actor_use_random_rotation_features, T_BOOL), K(actor_use_dirichlet_noise, T_BOOL), K(actor_dirichlet_noise_alpha, T_FLOAT),
    K(actor_dirichlet_noise_epsilon, T_FLOAT), K(actor_use_gumbel, T_BOOL), K(actor_use_gumbel_noise, T_BOOL),
    K(actor_gumbel_sample_size, T_INT), K(actor_gumbel_sigma_visit_c, T_FLOAT), K(actor_gumbel_sigma_scale_c, T_FLOAT),
    K(actor_resign_threshold, T_FLOAT), K(zero_num_threads, T_INT), K(zero_num_parallel_games, T_INT),
    K(zero_disable_resign_ratio, T_FLOAT), K(zero_actor_intermediate_sequence_length, T_INT), K(zero_actor_ignored_command, T_STRING),
    K(learner_muzero_unrolling_step, T_INT), K(learner_n_step_return, T_INT), K(zero_num_games_per_iteration, T_INT), K(zero_replay_buffer, T_INT),
    K(learner_use_per, T_BOOL), K(learner_per_alpha, T_FLOAT), K(learner_per_init_beta, T_FLOAT), K(learner_batch_size, T_INT), K(nn_file_name, T_STRING), K(nn_type_name, T_STRING),
    K(env_board_size, T_INT), K(env_go_komi, T_FLOAT), K(env_go_ko_rule, T_STRING), K(env_game, T_STRING), K(atari_init_q, T_BOOL), K(mz_pipeline_lanes, T_INT), K(mz_rng_streams, T_INT), K(mz_zero_copy, T_INT), K(mz_cpu_base, T_INT), K(mz_signal_wait, T_BOOL), K(mz_device_env, T_BOOL), K(mz_raw_observations, T_BOOL), K(mz_sim_kernel, T_BOOL), K(mz_sim_cluster, T_BOOL), K(mz_sim_split, T_BOOL), K(mz_sim_rounds, T_BOOL), K(mz_sim_round_min, T_INT), K(mz_sim_round_alt, T_BOOL), K(mz_sim_round_batch, T_BOOL), K(mz_sim_rounds_board, T_BOOL), K(mz_sim_round_pairs, T_BOOL), K(mz_sim_round_leaves, T_INT), K(mz_manual_step, T_BOOL), K(mz_nn_precision, T_STRING), K(env_atari_name, T_STRING), K(env_atari_episode_length, T_INT),
};
#undef K

// ref config/configuration.cpp:92-205: every other registered key
const char* const kPassiveKeys[] = {
    "program_use_color_message", "zero_server_port", "zero_training_directory", "zero_start_iteration",
    "zero_end_iteration", "zero_server_accept_different_model_games", "zero_display_latest_games", 
    "learner_per_beta_anneal", "learner_training_step", "learner_training_display_step",
    "learner_optimizer", "learner_learning_rate", "learner_momentum", "learner_weight_decay", "learner_value_loss_scale",
    "learner_num_thread", "nn_num_blocks", "nn_num_hidden_channels", "nn_num_value_hidden_channels", "env_atari_rom_dir",
    "env_conhex_use_swap_rule", "env_gomoku_rule", "env_gomoku_exactly_five_stones", "env_havannah_use_swap_rule", "env_hex_use_swap_rule",
    "env_killallgo_ko_rule", "env_killallgo_use_seki", "env_rubiks_scramble_rotate", "env_surakarta_no_capture_plies",
    "env_tetris_block_puzzle_num_holding_block", "env_tetris_block_puzzle_num_preview_holding_block",
};

std::string trimmed(const std::string& s)
{
    const size_t b = s.find_first_not_of(" \t");
    if (b == std::string::npos) { return ""; }
    return s.substr(b, s.find_last_not_of(" \t") - b + 1);
}

template <class T>
bool extract(const std::string& v, T* out) // stream extraction with nothing left over (ref configure_loader.h:12-17)
{
    std::istringstream iss(v);
    iss >> *out;
    return iss && iss.rdbuf()->in_avail() == 0;
}
} // namespace

bool WorkerConfig::loadFromString(const std::string& s)
{
    if (s.empty()) { setError("empty configuration string"); return false; }
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t end = s.find(':', pos);
        if (end == std::string::npos) { end = s.size(); }
        std::string item = s.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty() || item[0] == '#') { continue; }
        const size_t eq = item.find('=');
        std::string key = trimmed(item.substr(0, eq));
        std::string value = eq == std::string::npos ? item : item.substr(eq + 1);
        if (value.find('#') != std::string::npos) { value = value.substr(0, value.find('#')); }
        value = trimmed(value);
        const Key* k = nullptr;
        for (const Key& c : kKeys) { if (key == c.name) { k = &c; break; } }
        if (!k) {
            // keys of the reference's configuration that this path never reads (server, learner, other games) are
            // accepted so that a real minizero .cfg file loads unchanged
            bool passive = false;
            for (const char* pk : kPassiveKeys) { if (key == pk) { passive = true; break; } }
            if (passive) { continue; }
            setError("Invalid key \"%s\" and value \"%s\"", key.c_str(), value.c_str());
            return false;
        }
        char* field = reinterpret_cast<char*>(this) + k->offset;
        bool ok = true;
        switch (k->type) {
            case T_INT: ok = extract(value, reinterpret_cast<int*>(field)); break;
            case T_FLOAT: ok = extract(value, reinterpret_cast<float*>(field)); break;
            case T_CHAR: ok = extract(value, field); break;
            case T_STRING: *reinterpret_cast<std::string*>(field) = value; break;
            case T_BOOL: {
                std::string u = value;
                std::transform(u.begin(), u.end(), u.begin(), ::toupper);
                ok = (u == "TRUE" || u == "1" || u == "FALSE" || u == "0");
                if (ok) { *reinterpret_cast<bool*>(field) = (u == "TRUE" || u == "1"); }
                break;
            }
        }
        if (!ok) { setError("Unsatisfiable value \"%s\" for option \"%s\"", value.c_str(), key.c_str()); return false; }
    }
    return true;
}

} // namespace mz
