// Run-time configuration of the worker: the reference's `config::*` globals that the self-play path reads
// (ref config/configuration.cpp:7-90; same key names, defaults and "k=v:k=v" syntax as
// config/configure_loader.cpp:51-117), held in a struct so several workers can live in one process.
#pragma once
#include <string>

namespace mz {

struct WorkerConfig {
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
    int mz_rng_streams = 1;     // worker-only: host RNG streams; 1 = slave thread 0's generator for every game (deterministic contract), 0 = one per slave thread (zero_num_threads, ref actor_group.cpp:66-70), games in contiguous blocks
    int zero_num_parallel_games = 32;
    float zero_disable_resign_ratio = 0.1;
    int zero_actor_intermediate_sequence_length = 0;
    std::string zero_actor_ignored_command = "reset_actors";
    int learner_muzero_unrolling_step = 5;
    int learner_n_step_return = 0;
    // learner-side sampler (loader.cpp; ref configuration.cpp:40-67)
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
    // run-time replacements of the reference's compile-time switches (-D<GAME>, #if ATARI in mcts.cpp:211)
    std::string env_game = "tictactoe";
    bool atari_init_q = false;
    std::string env_atari_name = "ms_pacman";
    int env_atari_episode_length = 1000; // synthetic Atari-shaped environment: steps per episode
    // not a reference key: number of software-pipelined lanes the games are split into (1 = no pipelining)
    int mz_pipeline_lanes = 0; // 0 = chosen by the worker: 1, or 2 for lock-step pools whose cycles are long enough to hide one lane's tree kernels under the other's convolutions (worker.cpp wantsTwoLanes)
    // not a reference key: kernels read/write the pinned host staging directly (no per-cycle memcpy operations)
    // not a reference key: >= 0 pins the worker's host threads to consecutive CPUs (NUMA node of the caller first) from this index
    int mz_cpu_base = -1;
    // not a reference key: wait for the GPU on a pinned completion word (spin) instead of hipStreamSynchronize
    bool mz_signal_wait = true;
    bool mz_sim_split = true;   // a move's simulation-kernel launch in up to three parts, so that the host's noise / rotation draws overlap the parts already running
    bool mz_sim_rounds = true;  // muzero_atari with a Gumbel root: the leaves of a whole Gumbel round (the simulations between two halvings visit different root children)
                                // are evaluated side by side ahead of the simulations that consume them in order (sim.hip sim_pre_kernel_mz); false: every simulation evaluates its own leaf
    bool mz_sim_round_alt = true; // ... and, where a round leaves half of the CUs idle, a second expected leaf per simulation
    bool mz_sim_rounds_board = true; // ... and for MuZero board games with a Gumbel root (one workgroup per leaf; pays where the pool leaves CUs idle)
    bool mz_sim_round_batch = true; // ... by the batched pipeline of sim_rounds.hip (false: one workgroup per leaf, sim.hip sim_pre_kernel_mz; same entries)
    bool mz_sim_round_pairs = true; // ... and a round that leaves half of the CUs idle and does not use its second leaves runs every trunk on two workgroups (sim.hip sim_pre_pair_kernel_mz)
    int mz_sim_round_leaves = 0;    // ... with this many leaves per trunk workgroup (1, 2, 4; 0 = as many as keep every CU busy)
    int mz_sim_round_min = 2;   // ... for the rounds of at least this many simulations (2 = every round of a 50-simulation, 16-sample search: 16, 8, 4, 4, 4, 2 x 7)
    bool mz_sim_cluster = true; // muzero_atari simulation kernel: four workgroups per game when 4 x games <= CUs (sim_cluster.h)
    bool mz_sim_kernel = true; // with mz_device_env: whole runs of cycles as one launch of the per-game simulation kernel (sim.hip)
    bool mz_raw_observations = true; // muzero_atari roots: ship the observation ring as bytes, expand the float planes on the device
    bool mz_device_env = true; // AlphaZero Go / Othello / TicTacToe: rules, planes, legal mask and candidate sort on the device (go_dev.hip)
    // not a reference key: the worker never plays on its own — run_cycles stops when the searches are complete and the caller decides
    // (mz_worker_search_action / mz_worker_act / mz_worker_reset_search): the per-actor stepping of BaseActor / ZeroActor::think()
    bool mz_manual_step = false;
    // not a reference key: arithmetic of the residual tower — "f32" (default: bit-exact against the oracle, records identical to the reference)
    // or "bf16x3" (opt-in: split-bf16 operands on the 16-bit MFMA, outputs within 1e-3, records not bit-identical; net_bf16_body.h)
    std::string mz_nn_precision = "f32";
    int mz_zero_copy = 3; // bit 0: kernels read their inputs from pinned host memory; bit 1: kernels write their outputs there

    // returns false (and sets the library error string) on an unknown key or an unparsable value,
    // like ConfigureLoader::loadFromString; keys are applied left to right, later ones win
    bool loadFromString(const std::string& s);
};

} // namespace mz
