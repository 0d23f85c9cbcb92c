// Where the facades take the worker's configuration from.
//   * inside the reference tree (config/configuration.h on the include path): from the reference's own `minizero::config::*` globals, which
//     mode_handler.cpp has filled from -conf_file / -conf_str (ref console/mode_handler.cpp:42-76, config/configuration.cpp:92-205) — so
//     `actor::ActorGroup ag; ag.run();` (mode_handler.cpp:145-149) compiles and runs unchanged;
//   * stand-alone (apps/mzgpu_sp.cpp, the facade tests): from the string holder below.
// The string is the reference's "k=v:k=v" syntax (config/configure_loader.cpp:51-117) plus env_game (the reference bakes the game in at
// compile time: -DGO, -DOTHELLO, -DTICTACTOE, -DATARI).
#pragma once
#include <sstream>
#include <string>

#if __has_include("configuration.h") && !defined(MZGPU_NO_REFERENCE_CONFIG)
#include "configuration.h"
#define MZGPU_HAVE_REFERENCE_CONFIG 1
#endif

namespace minizero::config {

inline std::string& mzgpuConfigurationString()
{
    static std::string s;
    return s;
}

// value of the LAST `key=` in a configuration string ("" if absent)
inline std::string mzgpuConfValue(const std::string& conf, const std::string& key)
{
    auto trim = [](const std::string& t) {
        const size_t b = t.find_first_not_of(" \t");
        return b == std::string::npos ? std::string() : t.substr(b, t.find_last_not_of(" \t") - b + 1);
    };
    std::string value;
    size_t pos = 0;
    while (pos <= conf.size()) {
        size_t end = conf.find(':', pos);
        if (end == std::string::npos) { end = conf.size(); }
        std::string item = conf.substr(pos, end - pos);
        pos = end + 1;
        const size_t eq = item.find('=');
        if (eq == std::string::npos || trim(item.substr(0, eq)) != key) { continue; }
        item = item.substr(eq + 1);
        if (item.find('#') != std::string::npos) { item = item.substr(0, item.find('#')); }
        value = trim(item);
    }
    return value;
}

// the merged configuration the worker is created with
inline std::string mzgpuCollectConfiguration()
{
    if (!mzgpuConfigurationString().empty()) { return mzgpuConfigurationString(); }
#ifdef MZGPU_HAVE_REFERENCE_CONFIG
    std::ostringstream o;
    o << std::boolalpha;
#define MZGPU_KEY(k) o << #k "=" << k << ":";
    MZGPU_KEY(program_seed) MZGPU_KEY(program_auto_seed) MZGPU_KEY(program_quiet) MZGPU_KEY(actor_num_simulation) MZGPU_KEY(actor_mcts_puct_base)
    MZGPU_KEY(actor_mcts_puct_init) MZGPU_KEY(actor_mcts_reward_discount) MZGPU_KEY(actor_mcts_think_batch_size) MZGPU_KEY(actor_mcts_think_time_limit)
    MZGPU_KEY(actor_mcts_value_rescale) MZGPU_KEY(actor_mcts_value_flipping_player) MZGPU_KEY(actor_select_action_by_count)
    MZGPU_KEY(actor_select_action_by_softmax_count) MZGPU_KEY(actor_select_action_softmax_temperature)
    MZGPU_KEY(actor_select_action_softmax_temperature_decay) MZGPU_KEY(actor_use_random_rotation_features) MZGPU_KEY(actor_use_dirichlet_noise)
    MZGPU_KEY(actor_dirichlet_noise_alpha) MZGPU_KEY(actor_dirichlet_noise_epsilon) MZGPU_KEY(actor_use_gumbel) MZGPU_KEY(actor_use_gumbel_noise)
    MZGPU_KEY(actor_gumbel_sample_size) MZGPU_KEY(actor_gumbel_sigma_visit_c) MZGPU_KEY(actor_gumbel_sigma_scale_c) MZGPU_KEY(actor_resign_threshold)
    MZGPU_KEY(zero_num_threads) MZGPU_KEY(zero_num_parallel_games) MZGPU_KEY(zero_disable_resign_ratio)
    MZGPU_KEY(zero_actor_intermediate_sequence_length) MZGPU_KEY(zero_actor_ignored_command) MZGPU_KEY(learner_muzero_unrolling_step)
    MZGPU_KEY(learner_n_step_return) MZGPU_KEY(nn_file_name) MZGPU_KEY(nn_type_name) MZGPU_KEY(env_board_size) MZGPU_KEY(env_go_komi)
    MZGPU_KEY(env_go_ko_rule) MZGPU_KEY(env_atari_name)
#undef MZGPU_KEY
#if defined(GO) && GO
    o << "env_game=go";
#elif defined(OTHELLO) && OTHELLO
    o << "env_game=othello";
#elif defined(ATARI) && ATARI
    o << "env_game=atari:atari_init_q=true"; // the #if ATARI init-Q rule of mcts.cpp:211-216
#else
    o << "env_game=tictactoe";
#endif
    return o.str();
#else
    return "";
#endif
}

} // namespace minizero::config
