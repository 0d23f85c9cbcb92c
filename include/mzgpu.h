/* mzgpu.h - C ABI of the MI355X-native MiniZero self-play worker (libmzgpu.so).
 *
 * This is the drop-in boundary for the reference's self-play hot path: the reference's C++ classes
 * minizero::network::{Network,AlphaZeroNetwork,MuZeroNetwork} and minizero::actor::{MCTS,ZeroActor,
 * ActorGroup} become thin facades over these entry points (see INTEGRATION.md and
 * the include/minizero/ facade headers).  Plain C types only: opaque handles owned by the library, caller-owned
 * buffers, 0 / negative return codes with a thread-local error string (mz_last_error), no exceptions
 * or STL across the ABI.  One host thread drives one handle; every handle owns one HIP stream.
 *
 * "ref" citations are file:line under /root/reference/minizero/.
 */
#ifndef MZGPU_H
#define MZGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZ_OK 0
#define MZ_ERR_ARG (-1)      /* bad argument / unsupported shape */
#define MZ_ERR_DEVICE (-2)   /* HIP error or no GPU: the product has NO CPU fallback */
#define MZ_ERR_STATE (-3)    /* call not valid in the current state */
#define MZ_ERR_CAPACITY (-4) /* node pool / buffer capacity exceeded */

/* where = where the caller's data buffers live */
#define MZ_HOST 0
#define MZ_DEVICE 1

const char* mz_last_error(void);
/* replaces torch::cuda::device_count() (ref actor/actor_group.cpp:152,170) */
int mz_device_count(void);
/* CPUs the worker's spin-wait host pool may use: affinity mask capped by the cgroup CPU quota (no reference equivalent; the
 * reference's zero_num_threads is taken at face value, ref actor/actor_group.cpp:172-177). zero_num_threads is clamped to this - 1. */
int mz_usable_cpus(void);

/* ------------------------------------------------------------------------------------------
 * Network.  Replaces Network::loadModel + getters (ref network/network.cpp:14-42,
 * network/network.h:15-56) and the TorchScript forward of alphazero_network.h:63-104 /
 * muzero_network.h:97-178.  The 12 hyper-parameters are the arguments of the reference's
 * create_network (ref network/py/create_network.py:6-18), in that order.
 * ------------------------------------------------------------------------------------------ */
typedef struct mz_net_desc {
    char game_name[64];
    int num_input_channels, input_channel_height, input_channel_width;
    int num_hidden_channels, hidden_channel_height, hidden_channel_width;
    int num_action_feature_channels, num_blocks, action_size, num_value_hidden_channels, discrete_value_size;
    int type; /* 0 alphazero, 1 muzero, 2 muzero_atari (ref get_type_name, alphazero_network.py:42-44) */
} mz_net_desc;

typedef struct mz_net mz_net;

/* number of f32 values in the flat weight blob: every floating-point tensor of the module's
 * state_dict() in order (conv/linear weight+bias, BN weight/bias/running_mean/running_var),
 * num_batches_tracked skipped.  BN is folded by the library at load. */
long mz_net_param_count(const mz_net_desc* desc);
/* deterministic synthetic weights (SplitMix64 counter stream; the "synthetic fixed-weight network"
 * of BASELINE.json).  out must hold mz_net_param_count() floats. */
int mz_net_generate_weights(const mz_net_desc* desc, uint64_t seed, float* out);

/* Network::loadModel's torch::jit::load (ref network/network.h:18-37), without LibTorch: reads the TorchScript archive the
 * reference's trainer writes (learner/train.py:127) and returns the 12 hyper-parameters and the flat weight blob described
 * above.  Call with weights_out == NULL to get *count_out; capacity = floats available in weights_out. */
int mz_net_read_pt(const char* path, mz_net_desc* desc_out, float* weights_out, size_t capacity, size_t* count_out);

/* The same for `path` as Network::loadModel is given it (ref network/network.h:18-37): the TorchScript archive if it exists, else — for
 * "x.pt" — the sibling flat blob "x.mzw" (magic "MZW1", mz_net_desc, uint64 count, f32 data: minizero_amd/export_weights.py). */
int mz_net_read_weight_file(const char* path, mz_net_desc* desc_out, float* weights_out, size_t capacity, size_t* count_out);

/* The same, read ONCE and kept: what a process that drives several devices hands to each of them instead of letting every network
 * re-read the file (the reference does the latter, actor/actor_group.cpp:227-232; SURVEY.md 8(e) asks for one read).  NULL + mz_last_error()
 * on failure.  mz_weight_file_reads(): how many weight files this process has opened and parsed so far (any entry point). */
typedef struct mz_weights mz_weights;
mz_weights* mz_weights_read(const char* path);
const mz_net_desc* mz_weights_desc(const mz_weights* w);
const float* mz_weights_data(const mz_weights* w);
size_t mz_weights_count(const mz_weights* w);
void mz_weights_free(mz_weights* w);
uint64_t mz_weight_file_reads(void);

/* createNetwork(file, gpu_id) (ref network/create_network.h:11-30): here the caller hands the parsed
 * blob; device must be a valid GPU ordinal (gpu_id == -1 / CPU is NOT supported: MZ_ERR_DEVICE). */
mz_net* mz_net_create(int device, const mz_net_desc* desc, const float* weights, size_t count);
/* load_model on a live network (ref actor/actor_group.cpp:227-232) */
int mz_net_reload(mz_net* net, const float* weights, size_t count);
void mz_net_destroy(mz_net* net);
/* opt-in arithmetic of the residual tower: 0 = f32 MFMA (default; bit-exact against the CPU oracle, records identical to the reference),
 * 1 = "bf16x3": split-bf16 operands on the 16-bit MFMA with f32 accumulation — network outputs within 1e-3 of the f32 path (the north star's
 * tolerance), about 2x the leaf evaluations per second, records NOT bit-identical.  AlphaZero networks with 64 hidden channels on 9x9 / 8x8
 * boards; MZ_ERR_ARG otherwise.  The worker takes the same switch as the configuration key mz_nn_precision=f32|bf16x3. */
int mz_net_set_precision(mz_net* net, int mode);
int mz_net_get_desc(const mz_net* net, mz_net_desc* out);

/* AlphaZeroNetwork::forward() (ref network/alphazero_network.h:63-104): features [B][C_in][H][W] f32
 * -> policy[B][A] (softmax), policy_logit[B][A], value[B].  Blocking. */
int mz_net_forward_az(mz_net* net, const float* features, int batch, float* policy, float* policy_logit, float* value, int where);
/* MuZeroNetwork::initialInference() (ref muzero_network.h:97-104, muzero_network.py:137-143):
 * hidden_state[B][C][h][w] is also returned */
int mz_net_initial(mz_net* net, const float* features, int batch, float* policy, float* policy_logit, float* value, float* hidden_state,
                   int where);
/* MuZeroNetwork::recurrentInference() (ref muzero_network.h:106-119, muzero_network.py:146-152):
 * hidden_in[B][C][h][w], action_plane[B][a][h][w] -> outputs + reward[B] (0 for board games) */
int mz_net_recurrent(mz_net* net, const float* hidden_in, const float* action_plane, int batch, float* policy, float* policy_logit,
                     float* value, float* reward, float* hidden_out, int where);
/* muzero_atari: value / reward are 601-bin categorical heads (ref muzero_network.h:157-174).  With MZ_HOST buffers the
 * library returns the decoded scalars; with MZ_DEVICE buffers it leaves the softmax expectation in the transformed space
 * h(x) = sign(x)(sqrt(|x|+1)-1) + 0.001x and the caller applies mz_invert_value (ref utils/utils.h:102-108). */
float mz_invert_value(float v);
/* its forward direction h(x), the scale of the learner's 601-bin value / reward targets (ref utils/utils.h:93-100, atari.h:115-116) */
float mz_transform_value(float v);
/* measurement hook for bench.py: runs `iters` forwards of batch B on resident synthetic inputs and
 * returns HIP-event times on the network's own stream: total ms per forward, and ms spent in the
 * 3x3-convolution kernels per forward (the dominant kernel; roofline numerator in DESIGN.md). */
int mz_net_time_forward(mz_net* net, int batch, int iters, float* ms_total, float* ms_conv3x3, double* conv_flops_per_forward);
/* average HIP-event duration of ONE launch of the dominant kernel (residual-tower conv3x3 C->C, fused
 * bias+ReLU) at batch B, with its algorithmic FLOPs and compulsory HBM bytes per launch */
int mz_net_time_tower_conv(mz_net* net, int batch, int iters, float* ms_per_launch, double* flops_per_launch, double* bytes_per_launch);

/* ------------------------------------------------------------------------------------------
 * Search pool: the structure-of-arrays node pool of `games` trees in HBM.  Replaces
 * Tree/MCTSNode/MCTS (ref actor/tree.h:32-122, actor/mcts.h:17-119, actor/mcts.cpp:20-228).
 * Node ids are 32-bit indices local to a game, root = 0, children of a node are one contiguous run.
 * ------------------------------------------------------------------------------------------ */
typedef struct mz_search_cfg {
    int num_simulation;          /* actor_num_simulation (sizes the PUCT tables and path buffers) */
    float puct_base, puct_init;  /* actor_mcts_puct_base / _init (ref configuration.cpp:14-15) */
    float reward_discount;       /* actor_mcts_reward_discount */
    int value_rescale;           /* actor_mcts_value_rescale (value-bound multiset, ref mcts.cpp:219-228) */
    int flipping_player;         /* charToPlayer(actor_mcts_value_flipping_player): 1 = 'B', 2 = 'W' */
    int atari_init_q;            /* the #if ATARI init-Q rule (ref mcts.cpp:211-216) as a run-time switch */
} mz_search_cfg;

typedef struct mz_pool mz_pool;

/* ref actor_group.cpp:183 (tree_node_size = (n+1)*A) and tree.h:64-69: nodes_per_game = 1 + tree_node_size */
mz_pool* mz_pool_create(int device, int games, int nodes_per_game, int action_size, const mz_search_cfg* cfg);
void mz_pool_destroy(mz_pool* pool);
/* MCTS::reset() + ZeroActor::resetSearch() (ref mcts.cpp:77-82, zero_actor.cpp:29-34) for the games
 * with mask[g] != 0 (NULL = all); root_player[g] = player of the root's (id -1) action */
int mz_pool_reset_search(mz_pool* pool, const uint8_t* mask, const int* root_player);
/* MCTS::select()/selectFromNode() for every game (ref mcts.cpp:139-149,181-217; gumbel_zero.cpp:83-85):
 * start_node NULL or start_node[g] <= 0: from the root; else path = root + PUCT path below that node.
 * Outputs (host): path_len[g], paths[g*max_depth + d] (node ids), path_action[g*max_depth + d]
 * (action ids; entry 0 is the root's -1).  max_depth = mz_pool_max_depth(). */
int mz_pool_select(mz_pool* pool, const int* start_node, int* path_len, int* paths, int* path_action);
int mz_pool_max_depth(const mz_pool* pool);
/* MCTS::expand() + MCTS::backup() on the paths of the last select (ref mcts.cpp:151-179).
 * cand_count[g] = k (0 = terminal leaf: backup only); candidates of game g at [g*A, g*A+k), already
 * in child order (the caller ran the reference's std::sort, zero_actor.cpp:225-227,241-243);
 * cand_player[g] = player of the new children's actions. */
int mz_pool_expand_backup(mz_pool* pool, const int* cand_count, const int* cand_action, const float* cand_policy, const float* cand_logit,
                          const int* cand_player, const float* value, const float* reward);
/* ZeroActor::addNoiseToNodeChildren() result (ref zero_actor.cpp:194-213): overwrite policy / logit /
 * noise of the root's children of the games with mask[g] != 0; arrays are [g*A + i], i = child order */
int mz_pool_root_set_noise(mz_pool* pool, const uint8_t* mask, const float* policy, const float* logit, const float* noise);
/* root statistics needed by the per-move host logic (move decision, resign test, P/V record strings,
 * Gumbel halving; ref mcts.cpp:84-137, gumbel_zero.cpp:9-137): per game n = num_children[g], and at
 * [g*A + i] action/count/mean/policy/logit/noise/value/reward of child i; root_* are [g] */
int mz_pool_root_read(mz_pool* pool, int* num_children, int* action, float* count, float* mean, float* policy, float* logit, float* noise,
                      float* value, float* reward, float* root_count, float* root_mean, float* root_value, float* bound_lo, float* bound_hi,
                      int* bound_size);
/* test / debug: copy the first n nodes of game g out of the SoA pool */
int mz_pool_read_nodes(mz_pool* pool, int game, int n, int* action, int* player, int* num_children, int* first_child, float* mean,
                       float* count, float* policy, float* logit, float* noise, float* value, float* reward);
int mz_pool_num_nodes(mz_pool* pool, int game);

/* ------------------------------------------------------------------------------------------
 * Worker: the `-mode sp` loop.  Replaces ActorGroup::run() and its stdin/stdout protocol
 * (ref actor/actor_group.cpp:136-252, console/mode_handler.cpp:145-149).
 * ------------------------------------------------------------------------------------------ */
typedef struct mz_worker mz_worker;

/* conf: the reference's "k=v:k=v" configuration string (ref config/configure_loader.cpp:51-117) plus
 * env_game=tictactoe|go|othello (the reference picks the game at compile time).  The worker owns a
 * pool of zero_num_parallel_games trees and a network on `device`. */
/* desc == NULL and weights == NULL: the network is read from the configuration's nn_file_name (mz_net_read_weight_file), as
 * ActorGroup::createNeuralNetworks does (ref actor_group.cpp:168-177). */
mz_worker* mz_worker_create(int device, const char* conf, const mz_net_desc* desc, const float* weights, size_t count);
/* BaseActor::setNetwork(std::shared_ptr<Network>) (ref actor/zero_actor.cpp:100-112, actor_group.cpp:183-187): the worker runs on the CALLER's
 * network instead of loading a second copy — `net` must live on `device` and outlive the worker; a later mz_net_reload on it is what the
 * worker's next search sees; the worker's own `load_model <path>` then only renames (EV tag).  One host thread drives the worker and the
 * network (they share the network's stream). */
mz_worker* mz_worker_create_shared(int device, const char* conf, mz_net* net);
void mz_worker_destroy(mz_worker* w);
/* one stdin line: start | stop | load_model <path> | update_config k=v:.. | reset_actors | quit | other (ignored)
 * (ref actor_group.cpp:200-252).  load_model <path> reads the file itself (mz_net_read_weight_file) and checks that its hyper-parameters
 * are those of the running network; weights staged with mz_worker_set_weights (in-memory callers) take precedence and are consumed.
 * update_config: keys that size device state at creation (actor_num_simulation, zero_num_parallel_games, the PUCT / Gumbel constants,
 * env_*, nn_type_name, mz_*) cannot change on a live worker: MZ_ERR_ARG with the key named, nothing applied.
 * Returns 1 for quit, 0, or a negative error. */
int mz_worker_command(mz_worker* w, const char* line);
int mz_worker_set_weights(mz_worker* w, const float* weights, size_t count);
/* `load_model <path>` for a caller that has already read the file (mz_weights_read — one read for all the devices of a process): the same
 * hyper-parameter check, reload and rename as the command, without opening anything.  With a shared network: rename only. */
int mz_worker_load_model(mz_worker* w, const char* path, const mz_net_desc* desc, const float* weights, size_t count);
/* run n lock-step cycles (one simulation of every game per cycle, ref actor_group.cpp:139-147);
 * returns the number of cycles actually run (0 while stopped) or a negative error.  Cycles of one call that need nothing from the host
 * run as ONE kernel launch: call with mz_worker_cycles_per_move() (= actor_num_simulation + 1) and poll commands between calls. */
int mz_worker_run_cycles(mz_worker* w, int n);
int mz_worker_cycles_per_move(const mz_worker* w);
/* pipeline lanes the pool is cut into (mz_pipeline_lanes; 0 = chosen by the worker: 1, or 2 for lock-step pools with long cycles) — an execution detail, never visible in a record */
int mz_worker_lanes(const mz_worker* w);
/* next pending stdout line ("SelfPlay ... #", ref actor_group.cpp:24-50); returns its length, 0 if none.  buf == NULL: the length of the
 * next line without popping it (Atari records carry their observations as hex and can be tens of megabytes); a buffer that is too small
 * gives MZ_ERR_CAPACITY and leaves the line queued.  mz_worker_peek_record / mz_worker_record answer buf == NULL the same way.
 * NEVER blocks: lines leave in the order the games finished, and while the front line's OBS tag (Atari: gzip + hex of megabytes, done by sleeping
 * helper threads beside the following moves) is not complete the answer is 0 = none yet.  A line whose compression failed is dropped and reported
 * once (MZ_ERR_STATE); the lines behind it keep draining. */
int mz_worker_pop_line(mz_worker* w, char* buf, int cap);
/* blocks until every queued line is complete (stop / quit / the end of a run: after it mz_worker_pop_line drains the queue); returns the number of queued lines */
int mz_worker_wait_lines(mz_worker* w);
/* test / monitoring access: the record of game `game` as it stands, unfinished games included (BaseActor::getRecord with no extra
 * tags, ref actor/base_actor.cpp:39-57); returns its length.  Does not disturb the search. */
int mz_worker_peek_record(mz_worker* w, int game, char* buf, int cap);
/* BaseActor::getRecord(tags) (ref base_actor.cpp:39-57): the same with extra tags (later tags of the same key win, like addTag) */
int mz_worker_record(mz_worker* w, int game, const char* const* keys, const char* const* values, int ntags, char* buf, int cap);

/* Per-actor stepping: the BaseActor / ZeroActor surface (ref actor/base_actor.h:16-55, zero_actor.h:24-70, create_actor.h:10-19) for callers
 * that drive single games themselves (ZeroActor::think(), the console).  With mz_manual_step=true in the configuration the worker never plays
 * on its own: mz_worker_run_cycles stops (returning the cycles it ran) when the searches are complete (isSearchDone) and holds the decision:
 *   mz_worker_search_action   getSearchAction() / isResign(): action id, its player, resign flag (ref zero_actor.h:39-40)
 *   mz_worker_act             BaseActor::act(Action): 1 = played (the P/V/R action info of the completed search is attached,
 *                             base_actor.cpp:22-30), 0 = illegal
 *   mz_worker_reset_search    ZeroActor::resetSearch() of every game (zero_actor.cpp:29-34): required before the next run_cycles
 *   mz_worker_reset_game      ZeroActor::reset() without the search part: new game + resign coin (zero_actor.cpp:23-27)
 *   mz_worker_emit_game       ThreadSharedData::outputGame(actor): queues the `SelfPlay ... #` line of game g (actor_group.cpp:24-50)
 *   mz_worker_env_query       what = 0 isTerminal, 1 getTurn, 2 getEvalScore(false), 3 getEvalScore(true), 4 number of actions, 5 getReward */
int mz_worker_search_done(const mz_worker* w);
int mz_worker_search_action(const mz_worker* w, int game, int* action_id, int* player, int* is_resign);
int mz_worker_act(mz_worker* w, int game, int action_id, int player);
int mz_worker_reset_search(mz_worker* w);
/* ZeroActor::think's early end (ref zero_actor.cpp:40-45: the time limit broke the loop, `if (!isSearchDone()) handleSearchDone()`): the decision from the
 * simulations run so far; needs the root's expansion (>= 2 cycles of the search).  After it mz_worker_search_done() == 1 and the held action is valid. */
int mz_worker_finish_search(mz_worker* w);
int mz_worker_reset_game(mz_worker* w, int game);
int mz_worker_emit_game(mz_worker* w, int game);
int mz_worker_env_query(const mz_worker* w, int game, int what, float* out);
/*   mz_worker_act_string             BaseActor::act(const std::vector<std::string>&) (ref base_actor.cpp:32-40): args = {player char, action string}
 *                                    ("B" "E5" / "w" "pass" for board games — ref utils/sgf_loader.cpp:89-99 —, an ALE action name such as "UPFIRE" for the
 *                                    Atari-shaped game, ref atari.cpp:9-39); 1 = played, 0 = not an action / illegal
 *   mz_worker_action_info_history    BaseActor::getActionInfoHistory() (ref base_actor.h:33-34): moves joined by 0x1e, inside a move key 0x1f value 0x1f ...;
 *                                    returns the length (buf == NULL: length only)
 *   mz_worker_env_features           Environment::getFeatures(rotation) of the game's position (ref base_env.h:88); returns the float count (out == NULL: count only)
 *   mz_worker_env_legal_mask         isLegalAction(a) for every action id, for the player to move (ref base_env.h:83-84); out[action_size]
 *   mz_worker_env_set_turn           Environment::setTurn (ref base_env.h:103)
 *   mz_worker_env_reset_seed         Environment::reset(seed) (ref atari.h:55) — games without a seed just reset
 *   mz_worker_env_rotate_action      Environment::getRotateAction(action_id, rotation) (ref base_env.h:99) */
int mz_worker_act_string(mz_worker* w, int game, const char* const* args, int nargs);
int mz_worker_action_info_history(mz_worker* w, int game, char* buf, int cap);
int mz_worker_env_features(const mz_worker* w, int game, int rotation, float* out, int capacity);
int mz_worker_env_legal_mask(const mz_worker* w, int game, uint8_t* out, int capacity);
int mz_worker_env_set_turn(mz_worker* w, int game, int player);
int mz_worker_env_reset_seed(mz_worker* w, int game, int seed);
int mz_worker_env_rotate_action(const mz_worker* w, int game, int action_id, int rotation);
typedef struct mz_worker_stats {
    uint64_t cycles, leaf_evals, moves, games;
    double ms_select, ms_env, ms_forward, ms_expand, ms_move, ms_total;
    uint64_t sim_launches; /* launches of the per-game simulation kernel (sim.hip); 0 = the lock-step kernels ran */
    uint64_t sim_cycles;   /* cycles that ran inside those launches */
    uint64_t pre_evals;    /* Gumbel rounds (mz_sim_rounds): leaves evaluated ahead of their simulations ... */
    uint64_t pre_hits;     /* ... and simulations that found their leaf among them (the rest evaluated their own) */
    uint64_t pre_alt_hits; /* ... of which: the leaf was the simulation's SECOND expected one (mz_sim_round_alt) */
    uint64_t pre_launches; /* rounds (counted in sim_launches too) whose leaves were evaluated ahead (sim_pre_kernel_mz, or the pipeline of sim_rounds.hip) */
    uint64_t pre_batch_launches; /* ... of which: by the batched pipeline (mz_sim_round_batch; walks | trunks | FC GEMMs | tails = 5 kernel launches each) */
    uint64_t pre_pair_launches;  /* ... of which: on pairs of workgroups per leaf (mz_sim_round_pairs, sim_pre_pair_kernel_mz; counted per network) */
} mz_worker_stats;
int mz_worker_get_stats(mz_worker* w, mz_worker_stats* out);
mz_net* mz_worker_net(mz_worker* w);

/* ------------------------------------------------------------------------------------------
 * Learner-side sampler.  Replaces learner::DataLoader and its pybind surface (ref learner/data_loader.h:75-93, data_loader.cpp:200-255,
 * learner/pybind.cpp:62-84): load_data_from_file / sample_data / update_priority, with ONE slave thread (thread 0 seeds program_seed + 0).
 * conf: the reference's configuration string (+ env_game).  The feature planes of a batch are produced on the GPU (a replay of every sampled
 * game on the device rules engine, or the stored Atari screens expanded), the targets on the host.
 * sample_data: buffers for learner_batch_size samples, laid out like the numpy arrays learner/train.py hands the reference
 *   features [B][C*H*W], action_features [B][U][a*h*w] (muzero), policy [B][(U+1)][A] ([B][A] alphazero), value [B][(U+1)][V] ([B][V]),
 *   reward [B][U][V] (muzero), loss_scale [B], sampled_index [B][2] = (game, position); U = learner_muzero_unrolling_step, V = 601 for the
 *   Atari-shaped game, else 1.  where = MZ_HOST or MZ_DEVICE (all seven buffers on that side; alphazero: action_features / reward may be NULL).
 * mz_loader_shape(what): 0 batch size, 1..5 floats per sample of features / action_features / policy / value / reward.
 * add_record returns 1 (loaded), 0 (the record does not load: skipped like data_loader.cpp:120-127) or a negative error.
 * ------------------------------------------------------------------------------------------ */
typedef struct mz_loader mz_loader;
mz_loader* mz_loader_create(int device, const char* conf);
void mz_loader_destroy(mz_loader* l);
int mz_loader_load_data_from_file(mz_loader* l, const char* file_name); /* returns the number of records loaded */
int mz_loader_add_record(mz_loader* l, const char* line);
int mz_loader_sample_data(mz_loader* l, float* features, float* action_features, float* policy, float* value, float* reward, float* loss_scale,
                          int* sampled_index, int where);
int mz_loader_update_priority(mz_loader* l, const int* sampled_index, const float* batch_values);
int mz_loader_num_data(const mz_loader* l);
int mz_loader_num_games(const mz_loader* l);
int mz_loader_shape(const mz_loader* l, int what);

/* utils::compressString (ref utils/utils.h:35-91): what the `OBS[...]` tag of an Atari record holds — the gzip member
 * (boost::iostreams::gzip_compressor defaults) of n bytes as lower-case hex, NUL-terminated; "" for n == 0.  Returns the hex length
 * (out == NULL: length only) or a negative error. */
long mz_compress_string(const void* data, size_t n, char* out, size_t capacity);

/* ------------------------------------------------------------------------------------------
 * Host-side environment access (rules engines the worker uses for AlphaZero leaves; exposed so the
 * parity tests can run the reference's env_test-style playout/replay check, ref
 * console/mode_handler.cpp:167-192).  No device work.
 * ------------------------------------------------------------------------------------------ */
typedef struct mz_env mz_env;
mz_env* mz_env_create(const char* conf);
void mz_env_destroy(mz_env* e);
void mz_env_reset(mz_env* e);
void mz_env_reset_seed(mz_env* e, int seed); /* envs that draw a seed on reset (ref atari.h:54) */
float mz_env_reward(const mz_env* e);
int mz_env_act(mz_env* e, int action_id, int player);
int mz_env_turn(const mz_env* e);
int mz_env_is_terminal(const mz_env* e);
float mz_env_eval_score(const mz_env* e, int is_resign);
int mz_env_policy_size(const mz_env* e);
int mz_env_feature_size(const mz_env* e);
int mz_env_legal_mask(const mz_env* e, uint8_t* out);
int mz_env_features(const mz_env* e, int rotation, float* out);
/* the action id of an action string as BaseEnv::act(const std::vector<std::string>&) reads its second argument (ref utils/sgf_loader.cpp:89-99
 * boardCoordinateStringToActionID for the board games — the value is NOT range-checked there either —, ref atari.cpp:9-39 for the Atari-shaped game: -1 = unknown) */
int mz_env_action_from_string(const mz_env* e, const char* action_string);
/* the same planes bit-packed (the device format): channel c = ceil(P/32) words, bit p%32 of word p/32 */
int mz_env_feature_bits(const mz_env* e, int rotation, uint32_t* out);

/* ------------------------------------------------------------------------------------------
 * Device-resident leaf environment (AlphaZero Go): test access.  The worker uses it when
 * mz_device_env=true (default) instead of replaying the path on a host copy of the root
 * environment (ref actor/zero_actor.cpp:55,79,215-252; environment/go/go.cpp:132-308).
 * mz_godev_playout plays actions[0..root_prefix) on the host engine (the root), then
 * actions[root_prefix..count) one node per move on the DEVICE engine and returns, for the
 * root (step 0) and after every device move, what the worker consumes: bit-packed planes under
 * rotation rots[step], legal mask, terminal flag, Tromp-Taylor result, player to move.
 * steps = count - root_prefix + 1; feat_out [steps][18*ceil(P/32)], legal_out [steps][P+1].
 * mz_sort_candidates orders n policies like the reference's std::sort (policy descending,
 * ref zero_actor.cpp:225-227) on the device, ties included: order_out[i] = index of the i-th.
 * mz_invert_values_device applies the 601-bin decode (ref utils/utils.h:102-108 invertValue) to n values with the device
 * function the simulation kernel uses for muzero_atari (mz_invert_value is the host function of the lock-step path).
 * ------------------------------------------------------------------------------------------ */
int mz_godev_playout(int device, int board_size, float komi, const int* actions, int count, int root_prefix, const int* rots,
                     uint32_t* feat_out, uint8_t* legal_out, int* terminal_out, float* eval_out, int* player_out);
/* the same for any game with a device engine ("go", "othello", "tictactoe"): feat_out [steps][channels*ceil(P/32)], legal_out [steps][actions] */
int mz_envdev_playout(int device, const char* game, int board_size, float komi, const int* actions, int count, int root_prefix, const int* rots,
                      uint32_t* feat_out, uint8_t* legal_out, int* terminal_out, float* eval_out, int* player_out);
int mz_sort_candidates(int device, const float* policy, int n, int* order_out);
int mz_invert_values_device(int device, const float* values, int n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* MZGPU_H */
