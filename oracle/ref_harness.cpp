// ORACLE — TEST INFRASTRUCTURE ONLY.  Thin C entry points over the REFERENCE's own sources, compiled in
// place from /root/reference (never copied): utils/random.{h,cpp}, utils/rotation.h,
// config/{configuration,configure_loader}.cpp.  Output goes to oracle/_ref/ (git-ignored).
// These are the only reference C++ files on the hot path that build in this image without Boost.
#include "configuration.h"
#include "random.h"
#include "rotation.h"
#include <cstring>
#include <sstream>
#include <string>

using namespace minizero;

extern "C" {

void mzref_rng_vector(int seed, int kind, int n, int k, float alpha, double* out)
{
    utils::Random::seed(seed);
    int i = 0;
    while (i < n) {
        if (kind == 0) { out[i++] = utils::Random::randInt(); }
        else if (kind == 1) { out[i++] = utils::Random::randReal(); }
        else {
            std::vector<float> v = (kind == 2 ? utils::Random::randDirichlet(alpha, k) : utils::Random::randGumbel(k));
            for (int j = 0; j < k && i < n; ++j) { out[i++] = v[j]; }
        }
    }
}

int mzref_rotate(int rotation, int pos, int board_size) { return utils::getPositionByRotating(static_cast<utils::Rotation>(rotation), pos, board_size); }
int mzref_reversed_rotation(int rotation) { return static_cast<int>(utils::reversed_rotation[rotation]); }

// same "key=value\n" dump as mzo_config_dump (oracle/o_capi.cpp)
int mzref_config_dump(const char* conf, char* buf, int cap)
{
    config::ConfigureLoader cl;
    config::setConfiguration(cl);
    if (conf && *conf && !cl.loadFromString(conf)) { return -1; }
    using namespace config;
    std::ostringstream o;
    o << "program_seed=" << program_seed << "\nprogram_auto_seed=" << program_auto_seed << "\nprogram_quiet=" << program_quiet
      << "\nactor_num_simulation=" << actor_num_simulation << "\nactor_mcts_puct_base=" << actor_mcts_puct_base
      << "\nactor_mcts_puct_init=" << actor_mcts_puct_init << "\nactor_mcts_reward_discount=" << actor_mcts_reward_discount
      << "\nactor_mcts_think_batch_size=" << actor_mcts_think_batch_size << "\nactor_mcts_think_time_limit=" << actor_mcts_think_time_limit
      << "\nactor_mcts_value_rescale=" << actor_mcts_value_rescale << "\nactor_mcts_value_flipping_player=" << actor_mcts_value_flipping_player
      << "\nactor_select_action_by_count=" << actor_select_action_by_count
      << "\nactor_select_action_by_softmax_count=" << actor_select_action_by_softmax_count
      << "\nactor_select_action_softmax_temperature=" << actor_select_action_softmax_temperature
      << "\nactor_select_action_softmax_temperature_decay=" << actor_select_action_softmax_temperature_decay
      << "\nactor_use_random_rotation_features=" << actor_use_random_rotation_features
      << "\nactor_use_dirichlet_noise=" << actor_use_dirichlet_noise << "\nactor_dirichlet_noise_alpha=" << actor_dirichlet_noise_alpha
      << "\nactor_dirichlet_noise_epsilon=" << actor_dirichlet_noise_epsilon << "\nactor_use_gumbel=" << actor_use_gumbel
      << "\nactor_use_gumbel_noise=" << actor_use_gumbel_noise << "\nactor_gumbel_sample_size=" << actor_gumbel_sample_size
      << "\nactor_gumbel_sigma_visit_c=" << actor_gumbel_sigma_visit_c << "\nactor_gumbel_sigma_scale_c=" << actor_gumbel_sigma_scale_c
      << "\nactor_resign_threshold=" << actor_resign_threshold << "\nzero_num_threads=" << zero_num_threads
      << "\nzero_num_parallel_games=" << zero_num_parallel_games << "\nzero_disable_resign_ratio=" << zero_disable_resign_ratio
      << "\nzero_actor_intermediate_sequence_length=" << zero_actor_intermediate_sequence_length
      << "\nzero_actor_ignored_command=" << zero_actor_ignored_command << "\nlearner_muzero_unrolling_step=" << learner_muzero_unrolling_step
      << "\nlearner_n_step_return=" << learner_n_step_return << "\nnn_file_name=" << nn_file_name << "\nnn_type_name=" << nn_type_name
      << "\nenv_board_size=" << env_board_size << "\nenv_go_komi=" << env_go_komi << "\nenv_go_ko_rule=" << env_go_ko_rule << "\n";
    std::string s = o.str();
    int n = static_cast<int>(s.size());
    if (buf && cap > 0) {
        int m = n < cap - 1 ? n : cap - 1;
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return n;
}

} // extern "C"
