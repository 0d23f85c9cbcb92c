// The self-play worker of libmzgpu: the `-mode sp` loop of the reference (ref actor/actor_group.cpp:136-252)
// re-designed around one device-resident search pool per GPU.
//
// Per lock-step cycle (one simulation of every game, ref actor_group.cpp:139-147 / SURVEY.md §3.1):
//   device : expand+backup kernel  ->  PUCT select kernel  ->  network forward (f32 MFMA)
//   host   : per-game candidate lists (legal filter, inverse rotation, the reference's std::sort), leaf
//            environments (copy of the root position + path replay) and feature planes — embarrassingly
//            parallel over games on a thread pool — plus the strictly serial, RNG-ordered per-move logic
//            (root noise, move decision, resign coin, record strings) on one thread.
// The order in which the single mt19937 stream is consumed is the reference's: for actor 0..B-1:
// [root noise] [move decision] [reset: resign coin] [rotation draw].
#include "config.h"
#include "env.h"
#include "net.h"
#include "pool.h"
#include "go_dev.h"
#include "gumbel.h"
#include "host_threads.h"
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <random>
#include <sstream>
#include <thread>
#include <unordered_map>

namespace mz {

namespace {

struct Cand { int action; float policy, logit; };

using ActionInfo = std::vector<std::pair<std::string, std::string>>;

struct Game {
    std::unique_ptr<GameEnv> env, leaf;
    bool enable_resign = true;
    int rot = 0;
    // leaf cache (AlphaZero): filled when the features are built, consumed when the network output arrives
    bool leaf_terminal = false;
    float leaf_eval = 0, leaf_reward = 0;
    int leaf_turn = 1, path_len = 1;
    std::vector<uint8_t> legal;
    // Gumbel root state (ref gumbel_zero.h:18-21); candidates are indices into the root's children
    std::vector<int> candidates;
    int sample_size = 0, simulation_budget = 0;
    int selected = -1; // index of the chosen root child
    uint64_t raw_seen = ~0ull; // rawSerial() of the observation block last sent to the device
    std::vector<ActionInfo> action_info_history;
};

// optional host-side phase trace (MZ_TRACE=1): prints per-sub-step averages to stderr when the worker is destroyed
struct PhaseTrace {
    bool on = getenv("MZ_TRACE") != nullptr;
    double acc[20] = {0}, lo[20] = {0}, hi[20] = {0};
    uint64_t cnt[20] = {0};
    const char* names[20] = {"p1.sync", "p1.cand", "p1.expand_launch", "p1.rootread", "p1.serial", "p1.noise_reset", "p1.select_launch", "p2.sync", "p2.leaf",
                             "p2.fwd_launch", "p1.observations", "p1.noise_up", "p1.reset", "p1.upload_roots", "root.flush", "sim.prep", "iter.root", "iter.presim", "iter.simwait", ""};
    void add(int i, double ms)
    {
        if (!cnt[i] || ms < lo[i]) { lo[i] = ms; }
        if (!cnt[i] || ms > hi[i]) { hi[i] = ms; }
        acc[i] += ms; ++cnt[i];
        if (on && i < 18 && ms > 1.5) { fprintf(stderr, "[mz trace] slow %s #%llu: %.3f ms\n", names[i], (unsigned long long)cnt[i], ms); }
    }
    ~PhaseTrace()
    {
        if (!on) { return; }
        for (int i = 0; i < 20; ++i) { if (cnt[i]) { fprintf(stderr, "[mz trace] %-18s calls %8llu  avg %8.4f ms  (min %8.4f, max %8.4f)  total %10.2f ms\n", names[i], (unsigned long long)cnt[i], acc[i] / cnt[i], lo[i], hi[i], acc[i]); } }
    }
};

inline double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::string escapeSGF(const std::string& s) // ref base_env.h:303-313
{
    std::string out;
    for (char c : s) {
        if (c == '(' || c == ')' || c == '[' || c == ']' || c == '\\') { out += '\\'; }
        out += c;
    }
    return out;
}

} // namespace

// Workers of THIS process per physical device (one-process-for-all-GPUs with MZ_DEVICE_MAP=0,0, tools/multi_worker.py, a facade user with several actor groups): kernels
// that assume a whole idle GPU — the pairs of workgroups per leaf of sim_pre_pair_kernel_mz, which spin on each other — are only used by a worker that is alone on
// its device.  (Another PROCESS on the device cannot be seen from here: there the bounded wait + pairTrouble() of adaptRounds is the net, or mz_sim_round_pairs=false.)
static std::atomic<int> g_workers_on_device[64];

class Worker {
public:
    ~Worker()
    {
        obs_.reset(); // the compressor's threads write into queued lines: they end before anything else goes
        if (counted_device_ >= 0) { g_workers_on_device[counted_device_].fetch_sub(1); }
    }
    int counted_device_ = -1;
    // shared != nullptr: the worker runs on the caller's network (BaseActor::setNetwork's shared_ptr, ref zero_actor.cpp:100-112) instead of its own copy
    int init(int device, const char* conf, const mz_net_desc& desc, const float* weights, size_t count, Net* shared = nullptr);
    int command(const std::string& line);
    int loadModel(const std::string& path, const mz_net_desc& file_desc, const float* weights, size_t count);
    int setWeights(const float* w, size_t n)
    {
        pending_weights_.assign(w, w + n);
        return MZ_OK;
    }
    int runCycles(int n);
    // mz_pipeline_lanes = 0 (automatic): a pool that plays in the lock-step mode with the device rules — a shape without a simulation-kernel instance — and whose
    // cycle is dominated by long convolution kernels runs faster as TWO lanes on two streams: one lane's single-wave tree kernels (select, leaf, candidates, expand:
    // ~0.3 ms per cycle with 255 CUs idle) run under the other lane's convolutions.  Measured (profiles/r06_lockstep_lanes.json): 19x19 6b x 128, 256 games, 331 GFLOP
    // per cycle: 64.1 -> 67.1 k leaf-evals/s; 13x13 6b x 96, 73 GFLOP per cycle: 182 -> 167 k (its convolutions at 128 samples no longer fill the chip) — hence the bound.
    bool wantsTwoLanes() const
    {
        if (cfg_.mz_pipeline_lanes != 0 || lanes_.size() != 1 || shared_net_ || !resident_ || sim_kernel_ || G_ < 64) { return false; }
        const double P = double(desc_.hidden_channel_height) * desc_.hidden_channel_width, C = desc_.num_hidden_channels;
        const double conv_flops = 2.0 * 9.0 * P * (double(desc_.num_input_channels) * C + 2.0 * desc_.num_blocks * C * C);
        return double(G_) * conv_flops >= 2.0e11;
    }
    int cyclesPerMove() const { return n_ + 1; }
    int numLanes() const { return static_cast<int>(lanes_.size()); }
    int popLine(char* buf, int cap);
    int waitLines();
    int peekRecord(int game, char* buf, int cap, const char* const* keys = nullptr, const char* const* values = nullptr, int ntags = 0);
    // per-actor stepping (mz_manual_step=true): BaseActor / ZeroActor surface (ref actor/base_actor.h:16-55, zero_actor.h:24-70)
    bool searchDone() const { return search_done_; }
    int searchAction(int g, int* action_id, int* player, int* resign) const;
    int actGame(int g, int action_id, int player);
    int resetSearchAll();
    int finishSearch();
    int resetGameAt(int g);
    int emitGame(int g);
    int envQuery(int g, int what, float* out) const;
    int actString(int g, const char* const* args, int nargs);
    int actionInfoHistory(int g, char* buf, int cap);
    int envFeatures(int g, int rotation, float* out, int capacity) const;
    int envLegalMask(int g, uint8_t* out, int capacity) const;
    int envSetTurn(int g, int player);
    int envResetSeed(int g, int seed);
    int envRotateAction(int g, int action_id, int rotation) const;
    mz_worker_stats stats_{};
    int getStats(mz_worker_stats* out)
    {
        *out = stats_;
        MZ_HIP(hipSetDevice(device_));
        for (auto& L : lanes_) {
            unsigned hits = 0, evals = 0, alt = 0;
            int rc = L->net->simPreStats(&hits, &evals, &alt);
            if (rc) { return rc; }
            out->pre_hits += hits; out->pre_evals += evals; out->pre_alt_hits += alt;
            out->pre_pair_launches += L->net->pre_pair_launches_;
        }
        return MZ_OK;
    }
    Net& net0() { return *lanes_[0]->net; }

private:
    // A lane = a contiguous slice of the games with its own device pool, network instance, HIP stream and staging.
    // Lanes are software-pipelined on the one host thread: while lane A's kernels (select / tower / heads) run, the
    // host builds candidates / leaf positions for lane B, and the two streams let the GPU overlap one lane's tower
    // (1 workgroup per sample: half the CUs at 128 games) with the other lane's tree kernels.  Results do not depend on
    // the lane count: the RNG-ordered serial section still visits the games in index order.
    struct Lane {
        int g0 = 0, n = 0;
        Net* net = nullptr;            // the lane's network: its own (own_net) or the caller's (mz_worker_create_shared)
        std::unique_ptr<Net> own_net;
        Pool pool;
        hipStream_t stream = nullptr;
        PinBuf<float> h_feat, h_out;
        PinBuf<uint8_t> h_raw;   // raw root observations (muzero_atari: bytes up, planes expanded on the device)
        DevBuf<uint8_t> d_raw;
        // incremental form: the device keeps the previous move's raw block (d_raw / d_raw2 alternate), the host sends the newest screen + 40 bytes per game
        PinBuf<uint8_t> h_new, h_meta;
        DevBuf<uint8_t> d_new, d_meta, d_raw2;
        int raw_cur = 0;          // which of (d_raw, d_raw2) holds the last uploaded / rebuilt block
        bool raw_have = false;    // ... and whether it holds one at all
        bool raw_incremental = false; // this root cycle: every game of the lane can be rebuilt from the previous block
        DevBuf<float> d_feat, d_out, d_hidden;
        Pool::View<float> h_policy, h_logit, h_value, h_reward, d_policy, d_logit, d_value, d_reward;
        DevBuf<int> d_src_idx, d_dst_idx, d_action_ids;
        int signal_seq = 0; // last completion signal queued on this lane's stream
        // device-resident leaf environment (AlphaZero Go): per-cycle rotations travel as a kernel argument
        GoDevice godev;
        RotPack rot{};
        int cycles_since_signal = 0;
        PinBuf<uint8_t> h_rot;   // per-game simulation kernel: rotations of a batch of cycles [cycle][game]
        DevBuf<uint8_t> d_rot;
        // MuZero simulation kernel: root planes (bit-packed), legal mask and player of every game, refreshed once per move
        PinBuf<uint32_t> h_rootfeat; DevBuf<uint32_t> d_rootfeat;
        PinBuf<unsigned long long> h_rootlegal; DevBuf<unsigned long long> d_rootlegal;
        PinBuf<int> h_rootturn; DevBuf<int> d_rootturn;
        bool rr_ahead = false;                // the root statistics of the finished search are already on their way (Pool::rootReadLaunch behind the move's last launch)
        PinBuf<unsigned> h_prestat;           // Gumbel rounds: the network's counters of leaves evaluated ahead, copied behind every move's launches (adaptRounds)
        PinBuf<int> h_gum; DevBuf<int> d_gum; // Gumbel state of every game (gumbel.h): the device runs the halving between simulations
        PinBuf<float> h_noise;   // Dirichlet noise of the root children drawn ahead of the launch [game][A]
        DevBuf<float> d_noise;
        // GPU time of the simulation-kernel launches (stats: ms_forward): one pair of events per part of a move's launch (runCyclesSim)
        static constexpr int kSimParts = 24; // parts a launch can be cut into (a Gumbel-round move of muzero_atari: one part per round that is evaluated ahead + the stretches between)
        int pre_epoch = 0;        // serial number of the current move's pre-evaluated leaves (mz_sim_rounds), 0: none
        hipEvent_t ev0[kSimParts] = {}, ev1[kSimParts] = {};
        hipStream_t up_stream = nullptr; // uploads of the draws for a later part while an earlier part runs on `stream`
        hipEvent_t ev_up = nullptr;
        int makeSimEvents()
        {
            for (int k = 0; k < kSimParts; ++k) { MZ_HIP(hipEventCreate(&ev0[k])); MZ_HIP(hipEventCreate(&ev1[k])); }
            MZ_HIP(hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking));
            MZ_HIP(hipEventCreateWithFlags(&ev_up, hipEventDisableTiming));
            return MZ_OK;
        }
        ~Lane()
        {
            for (int k = 0; k < kSimParts; ++k) { if (ev0[k]) { (void)hipEventDestroy(ev0[k]); } if (ev1[k]) { (void)hipEventDestroy(ev1[k]); } }
            if (ev_up) { (void)hipEventDestroy(ev_up); }
            if (up_stream) { (void)hipStreamDestroy(up_stream); }
        }
    };
    Lane& laneOf(int g) { return *lanes_[g / lane_size_ < int(lanes_.size()) ? g / lane_size_ : int(lanes_.size()) - 1]; }
    int phase1(Lane& L, bool root_expansion, bool done, bool launch_select = true);
    int phase2(Lane& L);
    int phase2Resident(Lane& L);
    int runCyclesSim(int n);
    void drawCycles(int b0, int b1, int noise_row);
    void drawStream(int t, int b0, int b1, int noise_row);
    int uploadRoots(Lane& L);
    int setupDeviceGumbel();
    int cycle();
    int createActors();
    int resetAllSearches();
    void resetGame(Game& g, Rng& rng);          // ZeroActor::reset (ref zero_actor.cpp:23-27)
    int rootPlayerFor(const Game& g) const { return g.env->numPlayers() == 2 ? 3 - g.env->turn() : g.env->turn(); }
    void buildCandidates(int g);
    void buildLeaf(int g);
    // per-move host logic on root statistics (rr_* mirrors)
    float normalizedMean(float reward, float mean, float count, int player, int g) const;
    int selectChildByMaxCount(int g) const;
    int selectChildBySoftmaxCount(int g, float temperature, float value_threshold = 0.1f);
    bool isResign(int g) const;
    std::string searchDistributionString(int g) const;
    std::string gumbelPolicyString(int g, int child_player) const;
    void gumbelSortByScore(int g);
    void gumbelSequentialHalving(int g);
    int decideAction(int g);
    void handleSearchDone(int g);
    void outputGame(Game& gm);
    std::pair<int, int> trainingDataRange(const Game& gm) const;
    // obs_raw != nullptr: the OBS tag is left as a one-byte placeholder and the observation bytes are returned for finishObservations()
    std::string record(const Game& gm, const ActionInfo& extra, std::string* obs_raw = nullptr) const;
    std::unique_ptr<ObsCompressor> obs_; // created with the first record that carries observations
    ActionInfo actionInfo(int g, int child_player) const;

    WorkerConfig cfg_;
    mz_net_desc desc_{};
    int device_ = 0, G_ = 0, A_ = 0, n_ = 0;
    std::vector<std::unique_ptr<Lane>> lanes_;
    int lane_size_ = 1;
    std::unique_ptr<ThreadPool> threads_;
    std::vector<Game> games_;
    Rng main_rng_;
    // The slave threads' generators (ref actor_group.cpp:66-70: thread id seeds program_seed + id).  mz_rng_streams = 1 (default): every game draws from the
    // generator of slave thread 0 — the deterministic contract, and one of the schedules the reference's first-come-first-served hand-out of actors to threads
    // (actor_group.cpp:18-22) can produce.  mz_rng_streams = 0: as many generators as the reference has, zero_num_threads = T, the games statically partitioned
    // over them in contiguous blocks (streamOf) — another of those schedules (mz_rng_streams = S > 1: S generators whatever zero_num_threads says).  Every draw of game g comes from rngOf(g) in the reference's per-actor order; draws of different
    // streams are independent, so the long runs of draws (root noise, rotations: drawCycles) are made by the streams side by side on the pool's threads.
    std::vector<Rng> rngs_;
    int streams_ = 1;
    // the pool's size: zero_num_threads, never more than the CPUs we may burn
    int hostThreads() const { return std::max(1, std::min(cfg_.zero_num_threads, usableCpus() - 1)); }
    int streamOf(int g) const { return static_cast<int>(static_cast<long long>(g) * streams_ / G_); }
    Rng& rngOf(int g) { return rngs_[streamOf(g)]; }
    std::vector<std::vector<float>> stream_noise_;   // per stream: scratch of a noise draw
    bool running_ = false, pending_ = false;
    int sims_done_ = 0; // root visit count of every game (lock-step: identical for all games)
    int sim_pre_ = 0, sim_post_ = 0; // its value before / after the expand+backup of the current cycle
    std::deque<std::unique_ptr<OutLine>> lines_;
    std::vector<float> pending_weights_;
    // root statistics mirrors (valid after rootRead)
    std::vector<int> rr_nc_, rr_action_, rr_bsize_;
    std::vector<float> rr_count_, rr_mean_, rr_policy_, rr_logit_, rr_noise_, rr_value_, rr_reward_, rr_root_count_, rr_root_mean_, rr_root_value_,
        rr_lo_, rr_hi_;
    std::vector<uint8_t> noise_mask_;
    std::vector<float> noise_policy_, noise_logit_, noise_noise_;
    PhaseTrace trace_;
    int flipping_player_ = 2;
    bool use_signal_ = true;  // wait on a pinned completion word written by a 1-thread kernel instead of hipStreamSynchronize
    bool feat_bits_ = false; // AlphaZero leaves travel host->device as bit-packed planes (all board-game planes are 0/1)
    bool resident_ = false;  // the whole cycle runs on the device (go_dev.hip): the host only does the RNG-ordered per-move logic
    struct DeferredInfo { int g, mover; size_t index; };
    // what the per-move host logic of one RNG stream's games leaves for the pool: with several streams (mz_rng_streams) the streams' blocks of games run side by
    // side on the pool's threads and their results are merged in stream order = game order (serialSection)
    struct StreamSink {
        uint64_t moves = 0, games = 0;
        std::vector<DeferredInfo> deferred;
        struct Line { std::unique_ptr<OutLine> line; std::string obs_raw; bool has_obs = false; };
        std::vector<Line> lines;
    };
    std::vector<StreamSink> sinks_;
    static thread_local StreamSink* tl_sink_; // the sink of the stream this thread is working for (nullptr: write through)
    void queueLine(std::unique_ptr<OutLine> line, std::string&& obs_raw, bool has_obs);
    void serialSection(Lane& L, bool want_noise, bool done, bool az);
    void serialGame(int g, bool want_noise, bool done, bool az, std::vector<float>& noise);
    std::vector<float> noise_scratch_;
    std::vector<DeferredInfo> deferred_; // record strings of the last move, built while the next launch runs
    bool defer_info_ = false;
    void flushDeferred();
    struct Held { int action = -1, player = 0; bool resign = false; ActionInfo info; };
    std::vector<Held> held_;  // mz_manual_step: the decision of the completed search of every game
    bool search_done_ = false, stop_now_ = false;
    GumbelView gum_{};        // constants of the device-side Gumbel step (state pointer set per lane)
    bool dev_gumbel_ = false; // Gumbel root logic inside the simulation kernel
    int raw_bytes_ = 0;             // > 0: root observations travel as bytes (GameEnv::rawFeatures) and are expanded on the device
    bool sim_root_host_ = false;    // muzero_atari on sim_kernel_mz: the root (96x96 representation) is evaluated by a lock-step cycle
    bool root_host_pending_ = false; // ... whose outputs the next phase1 still has to turn into the root's children (host candidate lists)
    int syncGumbel(Lane& L, bool to_device);
    bool sim_mz_ = false;     // MuZero board game on sim_kernel_mz (no device rules needed: the leaves have no environment)
    struct Round { int s0, R; float p_event = 0.0f; bool alt = false; }; // p_event: share of the recent moves in which a simulation of the round missed its leaf / took the second one
    std::vector<unsigned> prestat_prev_; // the counters as of the previous move (summed over the lanes)
    void adaptRounds();
    bool pairsUsable() // two workgroups per leaf (sim.hip sim_pre_pair_kernel_mz)
    {
        if (!(cfg_.mz_sim_round_pairs && lanes_.size() == 1 && net0().pairsAvailable())) { return false; }
        const int sharing = counted_device_ < 0 ? 1 : g_workers_on_device[counted_device_].load();
        if (sharing == 1) { return true; }
        if (!pairs_off_logged_) { // said once: otherwise nothing explains the slower rounds (an idle or not yet collected worker object on the device is enough)
            pairs_off_logged_ = true;
            fprintf(stderr, "[mzgpu] device %d carries %d workers of this process: sim_pre_pair_kernel_mz (two workgroups per leaf) stays off for this worker\n", device_, sharing);
        }
        return false;
    }
    bool pairs_off_logged_ = false;
    int slab_slots_ = 0;        // hidden-state slots per game
    std::vector<Round> rounds_; // mz_sim_rounds: the rounds of a move whose leaves are evaluated ahead (first simulation, size), from the Gumbel schedule of (n, m)
    void planRounds();
    SimMzMode sim_mode_;      // this worker's choice of MuZero simulation kernels (launch arguments: the network may be shared with other workers)
    bool shared_net_ = false; // the network belongs to the caller (mz_worker_create_shared): load_model only renames, the caller reloads
    bool sim_kernel_ = false; // ... and whole runs of cycles are ONE launch of the per-game simulation kernel (sim.hip)
};

// ------------------------------------------------------------------------------------------------
int Worker::init(int device, const char* conf, const mz_net_desc& desc, const float* weights, size_t count, Net* shared)
{
    if (!conf || !cfg_.loadFromString(conf)) { return MZ_ERR_ARG; }
    desc_ = desc;
    device_ = device;
    if (counted_device_ < 0 && device >= 0 && device < 64) { counted_device_ = device; g_workers_on_device[device].fetch_add(1); }
    if (cfg_.nn_type_name == "muzero" && desc.type == 0) { setError("nn_type_name=muzero but the network descriptor is alphazero"); return MZ_ERR_ARG; }
    if (cfg_.zero_num_parallel_games <= 0 || cfg_.actor_num_simulation <= 0) { setError("zero_num_parallel_games and actor_num_simulation must be > 0"); return MZ_ERR_ARG; }
    if (cfg_.zero_num_parallel_games > 4096) { setError("zero_num_parallel_games > 4096 (ref alphazero_network.h:120 kReserved_batch_size)"); return MZ_ERR_ARG; }
    G_ = cfg_.zero_num_parallel_games;
    A_ = desc.action_size;
    n_ = cfg_.actor_num_simulation;
    flipping_player_ = (cfg_.actor_mcts_value_flipping_player == 'B' || cfg_.actor_mcts_value_flipping_player == 'b') ? 1
                       : (cfg_.actor_mcts_value_flipping_player == 'W' || cfg_.actor_mcts_value_flipping_player == 'w') ? 2 : 3;
    mz_search_cfg sc{};
    sc.num_simulation = n_;
    sc.puct_base = cfg_.actor_mcts_puct_base;
    sc.puct_init = cfg_.actor_mcts_puct_init;
    sc.reward_discount = cfg_.actor_mcts_reward_discount;
    sc.value_rescale = cfg_.actor_mcts_value_rescale;
    sc.flipping_player = flipping_player_;
    sc.atari_init_q = cfg_.atari_init_q;
    int nl = std::max(1, cfg_.mz_pipeline_lanes); // (0 = automatic: one lane here; mz_worker_create asks wantsTwoLanes() afterwards)
    if (G_ < 2 * nl || shared) { nl = 1; } // a shared network has one stream: one lane
    shared_net_ = shared != nullptr;
    lane_size_ = (G_ + nl - 1) / nl;
    lanes_.clear();
    for (int l = 0; l < nl; ++l) {
        auto L = std::make_unique<Lane>();
        L->g0 = l * lane_size_;
        L->n = std::min(lane_size_, G_ - L->g0);
        int rc = MZ_OK;
        if (shared) {
            if (shared->device_ != device) { setError("worker: the shared network lives on device %d, the worker on %d", shared->device_, device); return MZ_ERR_ARG; }
            L->net = shared;
        } else {
            L->own_net = std::make_unique<Net>();
            L->net = L->own_net.get();
            if ((rc = L->net->init(device, desc, weights, count))) { return rc; }
        }
        if (cfg_.mz_nn_precision != "f32" && cfg_.mz_nn_precision != "bf16x3") { setError("mz_nn_precision '%s' unknown (f32 | bf16x3)", cfg_.mz_nn_precision.c_str()); return MZ_ERR_ARG; }
        // (a shared network keeps the precision its owner chose with mz_net_set_precision)
        if (!shared && (rc = L->net->setPrecision(cfg_.mz_nn_precision == "bf16x3" ? 1 : 0))) { return rc; }
        if (!cfg_.mz_sim_cluster || nl > 1) { sim_mode_.cluster = false; } // two lanes = two concurrent cooperative launches: not with clusters that wait for each other
        L->stream = L->net->stream_;
        // ref actor_group.cpp:183: tree_node_size = (n + 1) * action_size; tree.h:66: 1 + tree_node_size nodes
        rc = L->pool.init(device, L->n, 1 + (n_ + 1) * A_, A_, sc, L->stream);
        if (rc) { return rc; }
        // kernels talk to the pinned staging directly: no memcpy operations on the per-cycle path
        if (cfg_.mz_zero_copy & 1) { L->pool.zero_copy_ = true; }
        if (cfg_.mz_zero_copy & 2) {
            L->pool.v_.host_path_len = L->pool.h_path_len_.p;
            L->pool.v_.host_path_action = L->pool.h_path_action_.p;
        }
        const size_t GA = size_t(L->n) * A_, feat = size_t(L->n) * L->net->featSize(), Gn = L->n;
#define WALLOC(b, n) \
    if (!(b).alloc(n)) { setError("worker: allocation failed (%s)", #b); return MZ_ERR_DEVICE; }
        WALLOC(L->h_feat, feat); WALLOC(L->d_feat, feat); WALLOC(L->h_out, 2 * GA + 2 * Gn); WALLOC(L->d_out, 2 * GA + 2 * Gn);
        L->h_policy = {L->h_out.p, GA}; L->h_logit = {L->h_out.p + GA, GA}; L->h_value = {L->h_out.p + 2 * GA, Gn}; L->h_reward = {L->h_out.p + 2 * GA + Gn, Gn};
        L->d_policy = {L->d_out.p, GA}; L->d_logit = {L->d_out.p + GA, GA}; L->d_value = {L->d_out.p + 2 * GA, Gn}; L->d_reward = {L->d_out.p + 2 * GA + Gn, Gn};
        MZ_HIP(hipMemset(L->d_out.p, 0, L->d_out.n * sizeof(float)));
        if (desc.type >= 1) {
            // hidden-state slab: one slot per expanded node; with Gumbel rounds on muzero_atari a second bank for the rounds' second expected leaves
            slab_slots_ = (n_ + 1) * ((desc.type == 2 && cfg_.actor_use_gumbel && cfg_.mz_sim_rounds && cfg_.mz_sim_round_alt) ? 2 : 1);
            WALLOC(L->d_hidden, Gn * slab_slots_ * L->net->hiddenSize());
            WALLOC(L->d_src_idx, Gn); WALLOC(L->d_dst_idx, Gn); WALLOC(L->d_action_ids, Gn);
        }
#undef WALLOC
        lanes_.push_back(std::move(L));
    }
    // the pool spins: never more spinners than CPUs the container may use, minus one for the HIP runtime's helper threads
    threads_ = std::make_unique<ThreadPool>(hostThreads(), cfg_.mz_cpu_base);
    const size_t GA = size_t(G_) * A_;
    rr_nc_.resize(G_); rr_action_.resize(GA); rr_bsize_.resize(G_);
    for (auto* v : {&rr_count_, &rr_mean_, &rr_policy_, &rr_logit_, &rr_noise_, &rr_value_, &rr_reward_}) { v->resize(GA); }
    for (auto* v : {&rr_root_count_, &rr_root_mean_, &rr_root_value_, &rr_lo_, &rr_hi_}) { v->resize(G_); }
    noise_mask_.assign(G_, 1);
    held_.assign(G_, Held());
    noise_policy_.resize(GA); noise_logit_.resize(GA); noise_noise_.resize(GA);
    int rcc = createActors();
    if (rcc) { return rcc; }
    // AlphaZero board games: every plane is 0 / 1, so leaves reach the network bit-packed whatever kernels serve its shape — the fused towers stage the bits
    // themselves, the run-time-shaped per-layer kernels behind every other shape get them unpacked first (Net::runTrunk).  (Until round 6 this was tied to
    // hasFusedTower(), which also kept the device rules — resident_ below — away from every shape on conv3x3_band: a host hop per lock-step cycle.)
    feat_bits_ = (desc.type == 0);
    raw_bytes_ = (desc.type == 2 && cfg_.mz_raw_observations) ? games_[0].env->rawFeatureBytes() : 0;
    if (raw_bytes_ > 0) {
        for (auto& L : lanes_) {
            if (!L->h_raw.alloc(size_t(L->n) * raw_bytes_) || !L->d_raw.alloc(size_t(L->n) * raw_bytes_)) { setError("worker: allocation failed (raw observations)"); return MZ_ERR_DEVICE; }
            const int fb = games_[0].env->rawFrameBytes(), mb = raw_bytes_ - 8 * fb;
            if (fb > 0 && mb > 0 && mb <= 64 &&
                (!L->h_new.alloc(size_t(L->n) * fb) || !L->d_new.alloc(size_t(L->n) * fb) || !L->h_meta.alloc(size_t(L->n) * mb) || !L->d_meta.alloc(size_t(L->n) * mb) ||
                 !L->d_raw2.alloc(size_t(L->n) * raw_bytes_))) {
                setError("worker: allocation failed (raw observations)"); return MZ_ERR_DEVICE;
            }
        }
    }
    use_signal_ = cfg_.mz_signal_wait;
    resident_ = cfg_.mz_device_env && desc.type == 0 && feat_bits_ && games_[0].env->hasDeviceTwin() && lane_size_ <= kRotPackGames;
    if (resident_) {
        const GameEnv& e = *games_[0].env;
        const int* inv[8];
        const int* fwd[8];
        for (int r = 0; r < 8; ++r) { inv[r] = e.rot()->inv[r].data(); fwd[r] = e.rot()->fwd[r].data(); }
        for (auto& L : lanes_) {
            int rc = L->godev.init(device, L->n, e.boardSize(), cfg_.env_go_komi, A_, n_ + 1, L->pool.v_.max_depth, L->stream, inv, fwd, e.zobristKeys(),
                                   e.deviceKind(), e.turnKey());
            if (rc) { return rc; }
            L->pool.v_.host_path_len = nullptr; // nobody on the host reads the paths any more
            L->pool.v_.host_path_action = nullptr;
            if ((rc = uploadRoots(*L))) { return rc; }
        }
        sim_kernel_ = cfg_.mz_sim_kernel && net0().hasSimKernel(e.boardSize(), e.deviceKind(), cfg_.actor_num_simulation) &&
                      (!cfg_.actor_use_gumbel || (cfg_.actor_gumbel_sample_size >= 2 && cfg_.actor_gumbel_sample_size <= kGumbelMaxSample));
        { int rcg = setupDeviceGumbel(); if (rcg) { return rcg; } }
        defer_info_ = sim_kernel_ && !cfg_.mz_manual_step;
        if (sim_kernel_) {
            for (auto& L : lanes_) {
                if (!L->h_rot.alloc(size_t(n_ + 1) * L->n) || !L->d_rot.alloc(size_t(n_ + 1) * L->n)) { setError("worker: allocation failed (rot table)"); return MZ_ERR_DEVICE; }
                if (!L->h_noise.alloc(size_t(L->n) * A_) || !L->d_noise.alloc(size_t(L->n) * A_)) { setError("worker: allocation failed (noise)"); return MZ_ERR_DEVICE; }
                { int rce = L->makeSimEvents(); if (rce) { return rce; } }
            }
        }
    }
    sim_mz_ = !resident_ && cfg_.mz_sim_kernel && (desc.type == 1 || desc.type == 2) && net0().hasSimKernelMz(cfg_.actor_num_simulation) &&
              (!cfg_.actor_use_gumbel || (cfg_.actor_gumbel_sample_size >= 2 && cfg_.actor_gumbel_sample_size <= kGumbelMaxSample));
    if (sim_mz_) {
        sim_kernel_ = true;
        defer_info_ = !cfg_.mz_manual_step;
        sim_root_host_ = desc.type == 2;
        { int rcg = setupDeviceGumbel(); if (rcg) { return rcg; } }
        if (dev_gumbel_ && (desc.type == 2 || (desc.type == 1 && cfg_.mz_sim_rounds_board && net0().hasPreBoard())) && cfg_.mz_sim_rounds && cfg_.mz_sim_split) {
            planRounds();
            for (auto& L : lanes_) {
                if (!L->h_prestat.alloc(512)) { setError("worker: allocation failed (round counters)"); return MZ_ERR_DEVICE; }
                memset(L->h_prestat.p, 0, 512 * sizeof(unsigned));
            }
            int covered = 0;
            for (const Round& rd : rounds_) { covered += rd.R; }
            sim_mode_.rounds = !rounds_.empty();
            sim_mode_.alt_base = (!rounds_.empty() && slab_slots_ == 2 * (n_ + 1)) ? n_ + 1 : 0;
            // every simulation of a move has its leaf evaluated ahead: what is left for the simulation kernel is the tree work of one wave per game, which one
            // workgroup per game does with less overhead than a cluster of four (no command / result exchange, no cooperative launch): 616 -> 656 k leaf-evals/s
            if (covered == n_ && !getenv("MZ_ROUNDS_CLUSTER")) { sim_mode_.cluster = false; }
        }
        const int fw = sim_root_host_ ? 1 : games_[0].env->featureWords(), LW = (A_ + 63) / 64;
        for (auto& L : lanes_) {
            if (!L->h_rootfeat.alloc(size_t(L->n) * fw) || !L->d_rootfeat.alloc(size_t(L->n) * fw) || !L->h_rootlegal.alloc(size_t(L->n) * LW) ||
                !L->d_rootlegal.alloc(size_t(L->n) * LW) || !L->h_rootturn.alloc(L->n) || !L->d_rootturn.alloc(L->n) ||
                !L->h_noise.alloc(size_t(L->n) * A_) || !L->d_noise.alloc(size_t(L->n) * A_) || !L->h_rot.alloc(size_t(n_ + 1) * L->n) ||
                !L->d_rot.alloc(size_t(n_ + 1) * L->n)) {
                setError("worker: allocation failed (MuZero root staging)");
                return MZ_ERR_DEVICE;
            }
            { int rce = L->makeSimEvents(); if (rce) { return rce; } }
            int rc = uploadRoots(*L);
            if (rc) { return rc; }
        }
    }
    return MZ_OK;
}

// The visiting order of a Gumbel root is a function of the visit COUNTS alone (ref gumbel_zero.cpp:74-119: fewest visits first, halving when every candidate has
// reached the budget): with m sampled children the simulations of a move fall into rounds in which every remaining candidate is visited once — m, m/2, ... —
// e.g. n = 50, m = 16: 16, 8, 4, 4, 4, 2 x 7.  The rounds of >= kMinRound simulations are evaluated ahead (sim.hip sim_pre_kernel_mz); a root with fewer than m
// children follows another schedule: its entries simply do not match and its simulations evaluate their own leaves.
void Worker::planRounds()
{
    rounds_.clear();
    const int kMinRound = std::max(1, cfg_.mz_sim_round_min);
    const int n = cfg_.actor_num_simulation, m = cfg_.actor_gumbel_sample_size;
    if (m < 2 || m > kGumbelMaxSample || A_ < m) { return; }
    std::vector<int> cnt(m, 0);
    int ncand = m, sample = m;
    int budget = static_cast<int>(std::max(1.0, std::floor(n / (std::log2(m) * m))));
    int s = 1;
    while (s <= n) {
        if (s > 1) { // the step before simulation s (gumbel_zero.cpp:104-118)
            bool all = true;
            for (int i = 0; i < ncand; ++i) { if (cnt[i] < budget) { all = false; break; } }
            if (all) {
                const int next_budget = static_cast<int>(std::floor(n / (std::log2(m) * sample / 2)));
                if (next_budget > 0 && sample > 2) {
                    sample /= 2;
                    if (ncand > sample) { ncand = sample; } // (which ones survive does not matter here: their counts are equal)
                    budget = cnt[0] + next_budget;
                }
            }
        }
        // a round: every candidate with the minimum count once
        int mn = cnt[0];
        for (int i = 1; i < ncand; ++i) { mn = std::min(mn, cnt[i]); }
        int R = 0;
        for (int i = 0; i < ncand; ++i) { if (cnt[i] == mn) { ++cnt[i]; ++R; } }
        R = std::min(R, n - s + 1);
        if (R >= kMinRound) { rounds_.push_back({s, R}); rounds_.back().alt = !lanes_.empty() && 2 * G_ * R <= net0().cuCount(); } // (the second leaves from the first move on where they fit: adaptRounds takes them away where they are not used)
        s += R;
    }
}

// A simulation that does not find its leaf evaluated ahead evaluates it itself, on ONE CU, while the round's launch — every other game of the pool — waits: one
// miss costs the move a whole evaluation's latency (150 us on BASELINE configs[4], where the second round of four misses in almost every move: the earlier
// simulations of the round move the value bounds, and the fourth visit of a root child is where a close arg-max flips).  A second expected leaf per simulation
// (mz_sim_round_alt) removes those misses but doubles the round's evaluations, so the rounds that do not fit the chip twice get it only while they keep needing it:
// per round, the share of recent moves in which one of its simulations missed (alt off) or took its second leaf (alt on), with hysteresis.  Which leaves are
// evaluated ahead never changes a record (sim.hip simPreProbe); the batched pipeline (sim_rounds.hip) evaluates the doubled round with two leaves per workgroup.
void Worker::adaptRounds()
{
    unsigned now[512] = {0};
    for (auto& L : lanes_) {
        if (!L->h_prestat.p) { return; }
        for (int i = 0; i < 512; ++i) { now[i] += L->h_prestat.p[i]; }
    }
    // (sim_pre_pair_kernel_mz: a partner workgroup stayed out — the GPU is shared.  Looked at whatever the second-leaf settings are: with mz_sim_round_alt=false every
    // small round goes to the pairs, and a shared GPU would pay the partner's bounded wait in every round of every move)
    if (now[129] != 0) { for (auto& L : lanes_) { L->net->pairTrouble(); } }
    if (!cfg_.mz_sim_round_alt || sim_mode_.alt_base == 0) { return; }
    if (prestat_prev_.size() == 512) {
        const int cus = net0().cuCount();
        for (Round& rd : rounds_) {
            bool ev = false;
            for (int s = rd.s0; s < rd.s0 + rd.R && s < 126; ++s) { ev = ev || now[2 + s] != prestat_prev_[2 + s] || now[256 + s] != prestat_prev_[256 + s]; }
            rd.p_event = 0.875f * rd.p_event + (ev ? 0.125f : 0.0f);
            // what the second leaves cost the round (us on BASELINE configs[4]) against a miss's 150 us: a round that fits the chip twice gives up its pairs of
            // workgroups per leaf (~30; nothing where pairs are not available: always on), a larger one doubles its evaluations (~80, batched pipeline only)
            const bool small = 2 * G_ * rd.R <= cus;
            if (!small && !cfg_.mz_sim_round_batch) { rd.alt = false; continue; }
            const float cost = small ? (pairsUsable() ? 30.0f : 0.0f) : 80.0f;
            if (rd.p_event * 150.0f > 1.3f * cost || cost == 0.0f) { rd.alt = true; } else if (rd.p_event * 150.0f < 0.6f * cost) { rd.alt = false; }
        }
    }
    prestat_prev_.assign(now, now + 512);
}

int Worker::setupDeviceGumbel() // the constants of the device-side Gumbel step + its per-lane state buffers
{
    dev_gumbel_ = sim_kernel_ && cfg_.actor_use_gumbel;
    if (dev_gumbel_) { // the constants of gumbel_zero.cpp:101,110 in the host's double arithmetic
        const int m = cfg_.actor_gumbel_sample_size;
        gum_.sample_size = m;
        gum_.sigma_visit_c = cfg_.actor_gumbel_sigma_visit_c;
        gum_.sigma_scale_c = cfg_.actor_gumbel_sigma_scale_c;
        gum_.budget0 = static_cast<int>(std::max(1.0, std::floor(cfg_.actor_num_simulation / (std::log2(m) * m))));
        gum_.num_simulation = cfg_.actor_num_simulation;
        gum_.log2_m = std::log2(m);
        for (auto& L : lanes_) {
            const size_t n = size_t(L->n) * (3 + kGumbelMaxSample);
            if (!L->h_gum.alloc(n) || !L->d_gum.alloc(n)) { setError("worker: allocation failed (gumbel state)"); return MZ_ERR_DEVICE; }
            memset(L->h_gum.p, 0, n * sizeof(int));
            MZ_HIP(hipMemset(L->d_gum.p, 0, n * sizeof(int)));
        }
    }
    return MZ_OK;
}

int Worker::uploadRoots(Lane& L)
{
    const int g0 = L.g0;
    if (sim_mz_) { // MuZero: what the initial inference and the root expansion need from the host engine
        const int fw = sim_root_host_ ? 1 : games_[0].env->featureWords(), LW = (A_ + 63) / 64;
        threads_->parallelFor(L.n, [this, &L, g0, fw, LW](int j) {
            Game& gm = games_[g0 + j];
            if (!sim_root_host_) { gm.env->featureBits(0, L.h_rootfeat.p + size_t(j) * fw); }
            gm.env->legalMask(gm.legal.data());
            for (int w = 0; w < LW; ++w) { L.h_rootlegal.p[size_t(j) * LW + w] = 0; }
            for (int a = 0; a < A_; ++a) { if (gm.legal[a]) { L.h_rootlegal.p[size_t(j) * LW + (a >> 6)] |= 1ull << (a & 63); } }
            L.h_rootturn.p[j] = gm.env->turn();
        });
        MZ_HIP(hipMemcpyAsync(L.d_rootfeat.p, L.h_rootfeat.p, L.h_rootfeat.n * sizeof(uint32_t), hipMemcpyHostToDevice, L.stream));
        MZ_HIP(hipMemcpyAsync(L.d_rootlegal.p, L.h_rootlegal.p, L.h_rootlegal.n * sizeof(unsigned long long), hipMemcpyHostToDevice, L.stream));
        MZ_HIP(hipMemcpyAsync(L.d_rootturn.p, L.h_rootturn.p, L.h_rootturn.n * sizeof(int), hipMemcpyHostToDevice, L.stream));
        return MZ_OK;
    }
    threads_->parallelFor(L.n, [this, &L, g0](int j) { games_[g0 + j].env->exportDeviceRoot(L.godev.hostSnap(j)); });
    return L.godev.uploadRoots();
}

int Worker::createActors()
{
    // ref mode_handler.cpp:62 (main-thread seed) + actor_group.cpp:179-187 (createActors -> reset() draws the resign coin
    // from the MAIN thread's generator) + actor_group.cpp:66-70 (slave thread 0 seeds its own generator: seed + 0)
    const int seed = cfg_.program_auto_seed ? static_cast<int>(std::random_device()()) : cfg_.program_seed;
    main_rng_.seed(seed);
    games_.clear();
    games_.resize(G_);
    for (auto& g : games_) {
        // ref atari.cpp:87: observations kept for the OBS tag of the next record
        const size_t recent_obs = static_cast<size_t>(cfg_.zero_actor_intermediate_sequence_length == 0
                                                          ? 108000
                                                          : cfg_.zero_actor_intermediate_sequence_length + 8 + cfg_.learner_n_step_return + cfg_.learner_muzero_unrolling_step) + 1;
        g.env = createGameEnv(cfg_.env_game, cfg_.env_board_size, cfg_.env_go_komi, cfg_.env_atari_name, cfg_.env_atari_episode_length, cfg_.env_go_ko_rule,
                              recent_obs);
        if (!g.env) { return MZ_ERR_ARG; }
        if (g.env->policySize() != A_ || g.env->featureSize() != net0().featSize()) {
            setError("network (A=%d, features=%d) does not fit env %s (A=%d, features=%d)", A_, net0().featSize(), g.env->name().c_str(),
                     g.env->policySize(), g.env->featureSize());
            return MZ_ERR_ARG;
        }
        g.leaf = g.env->clone();
        g.legal.assign(A_, 0);
        if (g.env->needsSeed()) {
            (void)main_rng_.randInt();                 // the reference env's constructor already resets once (atari.h:47-50)
            g.env->resetSeed(main_rng_.randInt());     // BaseActor::reset -> env_.reset() (atari.h:54)
        } else {
            g.env->reset();
        }
        g.action_info_history.clear();
        g.enable_resign = (main_rng_.randReal() < cfg_.zero_disable_resign_ratio ? false : true);
    }
    streams_ = std::max(1, cfg_.mz_rng_streams == 0 ? cfg_.zero_num_threads : cfg_.mz_rng_streams);
    rngs_.assign(streams_, Rng());
    for (int t = 0; t < streams_; ++t) { rngs_[t].seed(cfg_.program_auto_seed ? static_cast<int>(std::random_device()()) : cfg_.program_seed + t); }
    stream_noise_.assign(streams_, {});
    sims_done_ = 0;
    pending_ = false;
    return resetAllSearches();
}

int Worker::resetAllSearches()
{
    for (auto& L : lanes_) {
        std::vector<int> rp(L->n);
        for (int j = 0; j < L->n; ++j) { rp[j] = rootPlayerFor(games_[L->g0 + j]); }
        int rc = L->pool.resetSearch(nullptr, rp.data());
        if (rc) { return rc; }
        if ((resident_ || sim_mz_) && (rc = uploadRoots(*L))) { return rc; }
    }
    return MZ_OK;
}

void Worker::resetGame(Game& g, Rng& rng)
{
    if (g.env->needsSeed()) { g.env->resetSeed(rng.randInt()); } else { g.env->reset(); }
    g.action_info_history.clear();
    g.enable_resign = (rng.randReal() < cfg_.zero_disable_resign_ratio ? false : true);
}

// ------------------------------------------------------------------------------------------------
// candidates: ref zero_actor.cpp:215-245
void Worker::buildCandidates(int gi)
{
    Game& g = games_[gi];
    Lane& L = laneOf(gi);
    const int j = gi - L.g0; // index inside the lane
    const size_t off = size_t(j) * A_;
    int* ca = L.pool.h_cand_action_.p + off;
    float* cp = L.pool.h_cand_policy_.p + off;
    float* cl = L.pool.h_cand_logit_.p + off;
    const float* policy = L.h_policy.p + off;
    const float* logit = L.h_logit.p + off;
    Cand tmp[512];
    std::vector<Cand> big;
    Cand* c = tmp;
    if (A_ > 512) { big.resize(A_); c = big.data(); }
    int k = 0;
    float value = L.h_value.p[j], reward = 0.0f;
    int player;
    if (desc_.type == 0) {
        player = g.leaf_turn;
        reward = g.leaf_reward;
        if (g.leaf_terminal) {
            value = g.leaf_eval; // ref zero_actor.cpp:85: backup(path, env_transition.getEvalScore(), reward)
        } else {
            const int* fwd = g.env->rot()->fwd[g.rot].data();
            for (int a = 0; a < A_; ++a) {
                if (!g.legal[a]) { continue; }
                c[k++] = Cand{a, policy[fwd[a]], logit[fwd[a]]};
            }
        }
    } else {
        const int depth = g.path_len - 1; // depth of the leaf; its children are moved by the player to move there
        player = (g.env->numPlayers() == 2 && (depth & 1)) ? 3 - g.env->turn() : g.env->turn();
        reward = L.h_reward.p[j];
        if (desc_.type == 2) { // 601-bin heads: the kernels return the softmax expectation, the double-libm inverse stays here
            value = invertValueHost(value);
            reward = invertValueHost(reward);
        }
        for (int a = 0; a < A_; ++a) {
            if (depth == 0 && !g.legal[a]) { continue; } // legality is only known (and checked) at the root (zero_actor.cpp:238)
            c[k++] = Cand{a, policy[a], logit[a]};
        }
    }
    std::sort(c, c + k, [](const Cand& l, const Cand& r) { return l.policy > r.policy; }); // the reference's (unstable) std::sort
    for (int i = 0; i < k; ++i) { ca[i] = c[i].action; cp[i] = c[i].policy; cl[i] = c[i].logit; }
    L.pool.h_cand_count_.p[j] = k;
    L.pool.h_cand_player_.p[j] = player;
    L.pool.h_value_.p[j] = value;
    L.pool.h_reward_.p[j] = reward;
}

// leaf environment + features: ref zero_actor.cpp:51-72, 247-252
void Worker::buildLeaf(int gi)
{
    Game& g = games_[gi];
    Lane& L = laneOf(gi);
    const int j = gi - L.g0;
    const int len = L.pool.h_path_len_.p[j];
    g.path_len = len;
    float* feat = L.h_feat.p + size_t(j) * L.net->featSize();
    if (desc_.type == 0) {
        const int* acts = L.pool.h_path_action_.p + size_t(j) * L.pool.v_.max_depth;
        g.leaf->copyFrom(*g.env);
        int p = g.env->turn();
        for (int d = 1; d < len; ++d) {
            g.leaf->actUnchecked(acts[d], p);
            if (g.env->numPlayers() == 2) { p = 3 - p; }
        }
        g.leaf_terminal = g.leaf->isTerminal();
        g.leaf_turn = g.leaf->turn();
        g.leaf_reward = g.leaf->reward();
        if (g.leaf_terminal) { g.leaf_eval = g.leaf->evalScore(false); }
        else { g.leaf->legalMask(g.legal.data()); }
        if (feat_bits_) { g.leaf->featureBits(g.rot, reinterpret_cast<uint32_t*>(L.h_feat.p) + size_t(j) * g.env->featureWords()); }
        else { g.leaf->features(g.rot, feat); }
    } else if (sims_done_ == 0) {
        if (raw_bytes_ > 0 && L.raw_incremental) {
            const int fb = g.env->rawFrameBytes(), mb = raw_bytes_ - 8 * fb;
            g.env->rawNewest(L.h_new.p + size_t(j) * fb, L.h_meta.p + size_t(j) * mb);
            g.raw_seen = g.env->rawSerial();
        } else if (raw_bytes_ > 0) {
            g.env->rawFeatures(L.h_raw.p + size_t(j) * raw_bytes_);
            g.raw_seen = g.env->rawSerial();
        } else { g.env->features(0, feat); }
        g.env->legalMask(g.legal.data());
    }
}

// ------------------------------------------------------------------------------------------------
// host logic on root statistics
float Worker::normalizedMean(float reward, float mean, float count, int player, int g) const // ref mcts.cpp:40-53
{
    float value = reward + cfg_.actor_mcts_reward_discount * mean;
    if (cfg_.actor_mcts_value_rescale) {
        if (rr_bsize_[g] < 2) { return 1.0f; }
        const float lo = rr_lo_[g], hi = rr_hi_[g];
        value = (value - lo) / (hi - lo);
        value = ::fmin(static_cast<double>(1), ::fmax(static_cast<double>(-1), static_cast<double>(2 * value - 1)));
    }
    value = (player == flipping_player_ ? -value : value);
    value = (value * count - 0.0f) / (count + 0.0f);
    return value;
}

int Worker::selectChildByMaxCount(int g) const // ref mcts.cpp:91-104
{
    const size_t off = size_t(g) * A_;
    float max_count = 0.0f;
    int selected = -1;
    for (int i = 0; i < rr_nc_[g]; ++i) {
        if (rr_count_[off + i] <= max_count) { continue; }
        max_count = rr_count_[off + i];
        selected = i;
    }
    return selected;
}

int Worker::selectChildBySoftmaxCount(int g, float temperature, float value_threshold) // ref mcts.cpp:106-124
{
    const size_t off = size_t(g) * A_;
    const int child_player = games_[g].env->turn();
    int selected = -1;
    const int best = selectChildByMaxCount(g);
    const float best_mean = normalizedMean(rr_reward_[off + best], rr_mean_[off + best], rr_count_[off + best], child_player, g);
    float sum = 0.0f;
    const float exponent = 1 / temperature;
    for (int i = 0; i < rr_nc_[g]; ++i) {
        // powf(x, 1.0f) == x bit for bit (tests/csrc/pow_one_check.cpp), and an unvisited child is skipped before its mean is looked at:
        // this loop runs over ~17 k children per move of BASELINE configs[1] between two launches
        const float count = exponent == 1.0f ? rr_count_[off + i] : std::pow(rr_count_[off + i], exponent);
        if (count == 0) { continue; }
        const float mean = normalizedMean(rr_reward_[off + i], rr_mean_[off + i], rr_count_[off + i], child_player, g);
        if (mean < best_mean - value_threshold) { continue; }
        sum += count;
        float rand = rngOf(g).randReal(sum);
        if (selected == -1 || rand < count) { selected = i; }
    }
    return selected;
}

bool Worker::isResign(int g) const // ref mcts.cpp:84-89, zero_actor.h:40
{
    const Game& gm = games_[g];
    if (!gm.enable_resign) { return false; }
    const size_t off = size_t(g) * A_;
    // the root's own reward field is only ever written by the first backup of the search with the root-evaluation
    // reward, which is 0 for every environment this worker supports
    const float root_win_rate = normalizedMean(0.0f, rr_root_mean_[g], rr_root_count_[g], rootPlayerFor(gm), g);
    const int s = gm.selected;
    const float action_win_rate = normalizedMean(rr_reward_[off + s], rr_mean_[off + s], rr_count_[off + s], gm.env->turn(), g);
    return (-root_win_rate < cfg_.actor_resign_threshold && action_win_rate < cfg_.actor_resign_threshold);
}

std::string Worker::searchDistributionString(int g) const // ref mcts.cpp:126-137
{
    const size_t off = size_t(g) * A_;
    std::ostringstream oss;
    bool first = true;
    for (int i = 0; i < rr_nc_[g]; ++i) {
        if (rr_count_[off + i] == 0) { continue; }
        oss << (first ? "" : ",") << rr_action_[off + i] << ":" << rr_count_[off + i];
        first = false;
    }
    return oss.str();
}

std::string Worker::gumbelPolicyString(int g, int child_player) const // ref gumbel_zero.cpp:9-58
{
    const size_t off = size_t(g) * A_;
    const int nc = rr_nc_[g];
    float pi_sum = 0.0f, q_sum = 0.0f;
    for (int i = 0; i < nc; ++i) {
        if (rr_count_[off + i] == 0) { continue; }
        float value = normalizedMean(rr_reward_[off + i], rr_mean_[off + i], rr_count_[off + i], child_player, g);
        pi_sum += rr_policy_[off + i];
        q_sum += rr_policy_[off + i] * value;
    }
    float value_pi = rr_root_value_[g];
    if (cfg_.actor_mcts_value_rescale) {
        if (rr_bsize_[g] < 2) {
            value_pi = 1.0f;
        } else {
            value_pi = (value_pi - rr_lo_[g]) / (rr_hi_[g] - rr_lo_[g]);
            value_pi = ::fmin(static_cast<double>(1), ::fmax(static_cast<double>(-1), static_cast<double>(2 * value_pi - 1)));
        }
    }
    value_pi = (child_player == flipping_player_ ? -value_pi : value_pi);
    float non_visited_node_value = 1.0 / (1 + cfg_.actor_num_simulation) * (value_pi + (cfg_.actor_num_simulation / pi_sum) * q_sum);
    std::unordered_map<int, float> new_logits; // its iteration order is part of the record format
    float max_logit = -std::numeric_limits<float>::max();
    float max_child_count = 0;
    for (int i = 0; i < nc; ++i) { max_child_count = ::fmax(static_cast<double>(max_child_count), static_cast<double>(rr_count_[off + i])); }
    for (int i = 0; i < nc; ++i) {
        float value = (rr_count_[off + i] == 0 ? non_visited_node_value
                                               : normalizedMean(rr_reward_[off + i], rr_mean_[off + i], rr_count_[off + i], child_player, g));
        float logit_without_noise = rr_logit_[off + i] - rr_noise_[off + i];
        float score = logit_without_noise + (cfg_.actor_gumbel_sigma_visit_c + max_child_count) * cfg_.actor_gumbel_sigma_scale_c * value;
        new_logits.insert({rr_action_[off + i], score});
        max_logit = ::fmax(static_cast<double>(max_logit), static_cast<double>(score));
    }
    std::ostringstream oss;
    for (auto& logit : new_logits) {
        logit.second = logit.second - max_logit;
        if (logit.second < -38) { continue; }
        oss << (oss.str().empty() ? "" : ",") << logit.first << ":" << ::exp(static_cast<double>(logit.second));
    }
    return oss.str();
}

void Worker::gumbelSortByScore(int g) // ref gumbel_zero.cpp:121-137
{
    const size_t off = size_t(g) * A_;
    const int child_player = games_[g].env->turn();
    float max_child_count = 0;
    for (int i = 0; i < rr_nc_[g]; ++i) { max_child_count = ::fmax(static_cast<double>(max_child_count), static_cast<double>(rr_count_[off + i])); }
    auto score = [&](int i) {
        float min_value = -std::numeric_limits<float>::max();
        float value = normalizedMean(rr_reward_[off + i], rr_mean_[off + i], rr_count_[off + i], child_player, g);
        float s = rr_logit_[off + i] + (cfg_.actor_gumbel_sigma_visit_c + max_child_count) * cfg_.actor_gumbel_sigma_scale_c * value;
        return (rr_count_[off + i] > 0 ? s : min_value);
    };
    std::sort(games_[g].candidates.begin(), games_[g].candidates.end(), [&](int l, int r) { return score(l) > score(r); });
}

void Worker::gumbelSequentialHalving(int g) // ref gumbel_zero.cpp:90-119
{
    Game& gm = games_[g];
    const size_t off = size_t(g) * A_;
    if (sim_post_ == 1) {
        gm.candidates.clear();
        for (int i = 0; i < rr_nc_[g]; ++i) { gm.candidates.push_back(i); }
        std::sort(gm.candidates.begin(), gm.candidates.end(), [&](int l, int r) { return rr_logit_[off + l] > rr_logit_[off + r]; });
        if (static_cast<int>(gm.candidates.size()) > cfg_.actor_gumbel_sample_size) { gm.candidates.resize(cfg_.actor_gumbel_sample_size); }
        gm.sample_size = cfg_.actor_gumbel_sample_size;
        gm.simulation_budget = std::max(1.0, std::floor(cfg_.actor_num_simulation / (std::log2(cfg_.actor_gumbel_sample_size) * gm.sample_size)));
    } else {
        bool all = true;
        for (int i : gm.candidates) {
            if (rr_count_[off + i] >= gm.simulation_budget) { continue; }
            all = false;
            break;
        }
        if (all) {
            int next_budget = std::floor(cfg_.actor_num_simulation / (std::log2(cfg_.actor_gumbel_sample_size) * gm.sample_size / 2));
            if (next_budget > 0 && gm.sample_size > 2) {
                gm.sample_size /= 2;
                gumbelSortByScore(g);
                if (static_cast<int>(gm.candidates.size()) > gm.sample_size) { gm.candidates.resize(gm.sample_size); }
                gm.simulation_budget = rr_count_[off + gm.candidates[0]] + next_budget;
            }
        }
    }
}

int Worker::decideAction(int g) // ref zero_actor.cpp:178-192, gumbel_zero.cpp:60-72
{
    if (cfg_.actor_use_gumbel && cfg_.actor_select_action_by_count) {
        gumbelSortByScore(g);
        return games_[g].candidates[0];
    }
    if (!cfg_.actor_use_gumbel && cfg_.actor_select_action_by_count) { return selectChildByMaxCount(g); }
    if (cfg_.actor_select_action_by_softmax_count) { return selectChildBySoftmaxCount(g, cfg_.actor_select_action_softmax_temperature); }
    return selectChildByMaxCount(g);
}

ActionInfo Worker::actionInfo(int g, int child_player) const // ref base_actor.cpp:59-66, zero_actor.h:50-51, zero_actor.cpp:121-126
{
    ActionInfo info;
    info.push_back({"P", cfg_.actor_use_gumbel ? gumbelPolicyString(g, child_player) : searchDistributionString(g)});
    info.push_back({"V", std::to_string(rr_root_mean_[g])});
    std::ostringstream oss;
    oss << games_[g].env->reward();
    info.push_back({"R", oss.str()});
    return info;
}

static const char kObsPlaceholder[] = "\x01OBS\x01";

std::string Worker::record(const Game& gm, const ActionInfo& extra, std::string* obs_raw) const // ref base_actor.cpp:39-57, base_env.h:207-233
{
    ActionInfo tags;
    auto addTag = [&](const std::string& k, const std::string& v) {
        for (auto& t : tags) { if (t.first == k) { t.second = v; return; } }
        tags.push_back({k, v});
    };
    addTag("GM", gm.env->name());
    addTag("RE", std::to_string(gm.env->evalScore(false)));
    if (!gm.env->hasObservations()) {
        addTag("OBS", ""); // compressString("") == "" (utils.h:37): board games keep no observations
    } else if (obs_raw) {
        obs_raw->clear();
        gm.env->appendObservations(obs_raw);
        addTag("OBS", kObsPlaceholder);
    } else { // ref base_env.h:216-220
        std::string raw, hex;
        gm.env->appendObservations(&raw);
        (void)compressToHex(reinterpret_cast<const uint8_t*>(raw.data()), raw.size(), &hex);
        addTag("OBS", hex);
    }
    for (auto& t : gm.env->loaderTags()) { addTag(t.first, t.second); }
    addTag("EV", cfg_.nn_file_name.substr(cfg_.nn_file_name.find_last_of('/') + 1));
    if (!gm.env->isTerminal()) { // unfinished game = resigned: the player to move loses (base_actor.cpp:49-54)
        std::ostringstream oss;
        oss << gm.env->evalScore(true);
        addTag("RE", oss.str());
    }
    for (auto& t : extra) { addTag(t.first, t.second); }
    std::ostringstream oss;
    oss << "(;";
    for (const auto& t : tags) { oss << t.first << "[" << escapeSGF(t.second) << "]"; }
    const auto& ids = gm.env->actionIds();
    const auto& pls = gm.env->actionPlayers();
    const std::vector<int>* lives = gm.env->livesHistory(); // ref atari.cpp:187-197: L[lives] on the action before which a life was lost
    int previous_lives = lives && !lives->empty() ? (*lives)[0] : 0;
    for (size_t i = 0; i < ids.size(); ++i) {
        oss << ";" << (pls[i] == 1 ? 'B' : 'W') << "[" << ids[i] << "]";
        bool lost = false;
        int now = previous_lives;
        if (lives && i < lives->size()) { now = (*lives)[i]; lost = now < previous_lives; previous_lives = now; }
        if (gm.action_info_history.size() > i) {
            for (const auto& info : gm.action_info_history[i]) {
                const bool is_l = lost && info.first == "L"; // VectorMap: an existing key keeps its place and takes the new value
                oss << info.first << "[" << escapeSGF(is_l ? std::to_string(now) : info.second) << "]";
                if (is_l) { lost = false; }
            }
        }
        if (lost) { oss << "L[" << now << "]"; }
    }
    oss << ")";
    return oss.str();
}

std::pair<int, int> Worker::trainingDataRange(const Game& gm) const // ref actor_group.cpp:52-64
{
    int game_length = static_cast<int>(gm.env->actionIds().size());
    int data_start = 0, data_end = game_length - 1;
    const int seq = cfg_.zero_actor_intermediate_sequence_length;
    if (seq > 0) {
        const int un = cfg_.learner_muzero_unrolling_step + cfg_.learner_n_step_return;
        const bool term = gm.env->isTerminal();
        data_end = std::max(0, (term ? data_end : data_end - un));
        data_start = std::max(0, (term ? data_end - data_end % seq : data_end + 1 - seq));
        if (term && (data_end % seq < un)) { data_start = std::max(0, data_start - seq); }
    }
    return {data_start, data_end};
}

void Worker::outputGame(Game& gm) // ref actor_group.cpp:24-50
{
    const int game_length = static_cast<int>(gm.env->actionIds().size());
    const std::pair<int, int> range = trainingDataRange(gm);
    const bool is_terminal = (cfg_.zero_actor_intermediate_sequence_length == 0 || gm.env->isTerminal());
    std::ostringstream oss;
    std::string obs_raw;
    oss << "SelfPlay " << (is_terminal ? "true" : "false") << " " << (range.second - range.first + 1) << " " << game_length << " "
        << gm.env->evalScore(!gm.env->isTerminal()) << " "
        << record(gm, {{"DLEN", std::to_string(range.first) + "-" + std::to_string(range.second)}}, &obs_raw) << " "
        << "#";
    if (!is_terminal) {
        // (i < size: a resignation before the first move of an intermediate-sequence game has range 0-0 and an EMPTY history — the reference indexes it anyway)
        for (int i = range.first; i <= range.second && i < static_cast<int>(gm.action_info_history.size()); ++i) { gm.action_info_history[i].clear(); gm.action_info_history[i].shrink_to_fit(); }
    }
    auto line = std::make_unique<OutLine>();
    line->text = oss.str();
    const bool has_obs = gm.env->hasObservations();
    if (is_terminal) { if (tl_sink_) { ++tl_sink_->games; } else { ++stats_.games; } }
    if (tl_sink_) { tl_sink_->lines.push_back({std::move(line), std::move(obs_raw), has_obs}); }
    else { queueLine(std::move(line), std::move(obs_raw), has_obs); }
}

void Worker::queueLine(std::unique_ptr<OutLine> line, std::string&& obs_raw, bool has_obs)
{
    lines_.push_back(std::move(line));
    if (has_obs) {
        // (helpers: the host threads the configuration grants beyond the caller's, at least one — they sleep when there is nothing to compress)
        if (!obs_) { obs_ = std::make_unique<ObsCompressor>(std::max(1, hostThreads() - 1)); }
        obs_->submit(lines_.back().get(), std::move(obs_raw), kObsPlaceholder, sizeof(kObsPlaceholder) - 1);
    }
}

void Worker::handleSearchDone(int g) // ref actor_group.cpp:116-134 + base_actor.cpp:22-30
{
    Game& gm = games_[g];
    const size_t off = size_t(g) * A_;
    const bool resign = isResign(g);
    if (cfg_.mz_manual_step) { // the caller plays: keep what ZeroActor::handleSearchDone leaves behind (zero_actor.cpp:128-176)
        Held& h = held_[g];
        h.action = rr_action_[off + gm.selected];
        h.player = gm.env->turn();
        h.resign = resign;
        h.info = rr_root_count_[g] > 0 ? actionInfo(g, h.player) : ActionInfo(); // ZeroActor::getActionInfo (zero_actor.cpp:114-119)
        return;
    }
    bool acted = false;
    int mover = 0;
    if (!resign) {
        const int action = rr_action_[off + gm.selected];
        mover = gm.env->turn(); // == the player of every root child
        if (gm.env->act(action, mover)) {
            gm.action_info_history.resize(gm.env->actionIds().size());
            acted = true;
        }
    }
    if (tl_sink_) { ++tl_sink_->moves; } else { ++stats_.moves; }
    const bool is_endgame = (resign || gm.env->isTerminal());
    const int game_length = static_cast<int>(gm.env->actionIds().size());
    const int seq = cfg_.zero_actor_intermediate_sequence_length;
    const bool intermediate = !is_endgame && seq > 0 && game_length >= seq &&
                              (game_length - cfg_.learner_n_step_return - cfg_.learner_muzero_unrolling_step) % seq == 0;
    if (acted) {
        // the P/V/R strings of this move (no RNG involved) are only needed when the game is printed: for every other game they are
        // built after the next launch has been queued, off the critical path (the root statistics stay valid until the next root read)
        if (defer_info_ && !is_endgame && !intermediate) { (tl_sink_ ? tl_sink_->deferred : deferred_).push_back({g, mover, gm.action_info_history.size() - 1}); }
        else { gm.action_info_history.back() = actionInfo(g, mover); }
    }
    if (is_endgame) {
        outputGame(gm);
        resetGame(gm, rngOf(g));
    } else if (intermediate) {
        outputGame(gm);
    }
}

void Worker::flushDeferred()
{
    if (deferred_.empty()) { return; }
    threads_->parallelFor(static_cast<int>(deferred_.size()), [this](int k) {
        const DeferredInfo& d = deferred_[k];
        games_[d.g].action_info_history[d.index] = actionInfo(d.g, d.mover);
    });
    deferred_.clear();
}

thread_local Worker::StreamSink* Worker::tl_sink_ = nullptr;

// the per-move host logic of one game, in the reference's order: [root noise][move decision + Gumbel bookkeeping + act / record / reset][rotation]
void Worker::serialGame(int g, bool want_noise, bool done, bool az, std::vector<float>& noise)
{
    Game& gm = games_[g];
    const size_t off = size_t(g) * A_;
    if (want_noise) { // ref zero_actor.cpp:194-213
        const int k = rr_nc_[g];
        if (cfg_.actor_use_dirichlet_noise) {
            const float epsilon = cfg_.actor_dirichlet_noise_epsilon;
            rngOf(g).dirichlet(cfg_.actor_dirichlet_noise_alpha, k, noise);
            for (int i = 0; i < k; ++i) {
                rr_noise_[off + i] = noise[i];
                rr_policy_[off + i] = (1 - epsilon) * rr_policy_[off + i] + epsilon * noise[i];
            }
        } else {
            rngOf(g).gumbel(k, noise);
            for (int i = 0; i < k; ++i) {
                rr_noise_[off + i] = noise[i];
                rr_logit_[off + i] = rr_logit_[off + i] + noise[i];
            }
        }
        memcpy(noise_policy_.data() + off, rr_policy_.data() + off, k * sizeof(float));
        memcpy(noise_logit_.data() + off, rr_logit_.data() + off, k * sizeof(float));
        memcpy(noise_noise_.data() + off, rr_noise_.data() + off, k * sizeof(float));
    }
    if (done) { gm.selected = decideAction(g); }                  // zero_actor.cpp:96 handleSearchDone()
    if (cfg_.actor_use_gumbel) { gumbelSequentialHalving(g); }     // zero_actor.cpp:97
    if (done) { handleSearchDone(g); }                             // actor_group.cpp:92
    if (az && !(done && cfg_.mz_manual_step)) { // beforeNNEvaluation: the rotation draw (zero_actor.cpp:56)
        gm.rot = cfg_.actor_use_random_rotation_features ? rngOf(g).randInt() % 8 : 0;
    }
}

// One stream: strictly serial, in actor index order (the deterministic contract).  Several streams (mz_rng_streams): every stream's block of games in index
// order on one thread of the pool, the blocks side by side; what they produce for the pool (finished records, counters, deferred record strings) is merged in
// stream order, which is game order — the lines leave exactly as the serial loop would have queued them.
void Worker::serialSection(Lane& L, bool want_noise, bool done, bool az)
{
    const int g0 = L.g0, g1 = L.g0 + L.n;
    if (streams_ == 1) {
        std::vector<float>& noise = stream_noise_[0];
        for (int g = g0; g < g1; ++g) { serialGame(g, want_noise, done, az, noise); }
        return;
    }
    sinks_.resize(streams_);
    threads_->parallelFor(streams_, [this, g0, g1, want_noise, done, az](int t) {
        const int gs = std::max(g0, static_cast<int>((static_cast<long long>(t) * G_ + streams_ - 1) / streams_));
        const int ge = std::min(g1, static_cast<int>((static_cast<long long>(t + 1) * G_ + streams_ - 1) / streams_));
        if (gs >= ge) { return; }
        tl_sink_ = &sinks_[t];
        for (int g = gs; g < ge; ++g) { serialGame(g, want_noise, done, az, stream_noise_[t]); }
        tl_sink_ = nullptr;
    });
    for (StreamSink& s : sinks_) {
        stats_.moves += s.moves; stats_.games += s.games;
        s.moves = s.games = 0;
        deferred_.insert(deferred_.end(), s.deferred.begin(), s.deferred.end());
        s.deferred.clear();
        for (auto& l : s.lines) { queueLine(std::move(l.line), std::move(l.obs_raw), l.has_obs); }
        s.lines.clear();
    }
}

// ------------------------------------------------------------------------------------------------
// phase 1 of a lane: consume the network outputs of the previous cycle (expand + backup), run the RNG-ordered serial
// section for the lane's games, then launch the selection of the next simulation (asynchronously).
int Worker::phase1(Lane& L, bool root_expansion, bool done, bool launch_select)
{
    const bool az = desc_.type == 0;
    const int g0 = L.g0, g1 = L.g0 + L.n;
    double t0 = nowMs();
    if (pending_) {
        double t1 = t0, te = t0;
        int rc = MZ_OK;
        if (!resident_ && (!sim_mz_ || root_host_pending_)) { // resident / simulation kernel: candidates + expand + backup already ran on the device
            // (root_host_pending_ is cleared by the caller when EVERY lane has had its turn: cleared here, the lanes behind the first kept unexpanded roots)
            if (use_signal_) { int rcw = L.pool.waitSignal(L.signal_seq); if (rcw) { return rcw; } }
            else { MZ_HIP(hipStreamSynchronize(L.stream)); } // network outputs of this lane
            t1 = nowMs();
            stats_.ms_forward += t1 - t0;
            trace_.add(0, t1 - t0);
            threads_->parallelFor(L.n, [this, g0](int j) { buildCandidates(g0 + j); });
            const double tc = nowMs();
            trace_.add(1, tc - t1);
            rc = L.pool.expandBackupStaged(az ? -1 : sim_pre_);
            if (rc) { return rc; }
            te = nowMs();
            trace_.add(2, te - tc);
        }
        const bool want_noise = root_expansion && (cfg_.actor_use_dirichlet_noise || cfg_.actor_use_gumbel_noise);
        if (done || cfg_.actor_use_gumbel || want_noise) {
            // root statistics for the per-move host logic (sync point; also surfaces pool capacity errors)
            const size_t o = size_t(g0) * A_;
            rc = L.pool.rootRead(rr_nc_.data() + g0, rr_action_.data() + o, rr_count_.data() + o, rr_mean_.data() + o, rr_policy_.data() + o,
                                 rr_logit_.data() + o, rr_noise_.data() + o, rr_value_.data() + o, rr_reward_.data() + o, rr_root_count_.data() + g0,
                                 rr_root_mean_.data() + g0, rr_root_value_.data() + g0, rr_lo_.data() + g0, rr_hi_.data() + g0, rr_bsize_.data() + g0, L.rr_ahead);
            L.rr_ahead = false;
            if (rc) { return rc; }
            if ((rc = L.pool.checkError())) { return rc; }
        }
        const double t2 = nowMs();
        stats_.ms_expand += t2 - t1;
        trace_.add(3, t2 - te);
        // ---- the RNG-ordered section: actor index order within every RNG stream ----
        serialSection(L, want_noise, done, az);
        const double ts = nowMs();
        trace_.add(4, ts - t2);
        if (want_noise) {
            const size_t o = size_t(g0) * A_;
            if ((rc = L.pool.rootSetNoise(noise_mask_.data() + g0, noise_policy_.data() + o, noise_logit_.data() + o, noise_noise_.data() + o))) { return rc; }
            trace_.add(11, nowMs() - ts);
        }
        if (done && !cfg_.mz_manual_step) {
            std::vector<int> rp(L.n);
            for (int j = 0; j < L.n; ++j) { rp[j] = rootPlayerFor(games_[g0 + j]); }
            const double tr0 = nowMs();
            if ((rc = L.pool.resetSearch(nullptr, rp.data()))) { return rc; }
            const double tr1 = nowMs();
            trace_.add(12, tr1 - tr0);
            if ((resident_ || sim_mz_) && (rc = uploadRoots(L))) { return rc; }
            trace_.add(13, nowMs() - tr1);
        }
        if (done && cfg_.mz_manual_step) { return MZ_OK; } // no next selection: the caller acts and resets the search first
        t0 = nowMs();
        stats_.ms_move += t0 - t2;
        trace_.add(5, t0 - ts);
    } else {
        for (int g = g0; g < g1; ++g) {
            if (az) { games_[g].rot = cfg_.actor_use_random_rotation_features ? rngOf(g).randInt() % 8 : 0; }
        }
    }
    // ---- selection of the next simulation ----
    const int next_sim = pending_ ? (done ? 0 : sim_post_) : 0;
    const int* d_start = nullptr;
    if (cfg_.actor_use_gumbel && next_sim >= 1) { // ref gumbel_zero.cpp:74-88
        for (int g = g0; g < g1; ++g) {
            Game& gm = games_[g];
            const size_t off = size_t(g) * A_;
            std::sort(gm.candidates.begin(), gm.candidates.end(), [&](int l, int r) {
                return (rr_count_[off + l] < rr_count_[off + r] || (rr_count_[off + l] == rr_count_[off + r] && rr_logit_[off + l] > rr_logit_[off + r]));
            });
            L.pool.h_start_.p[g - g0] = 1 + gm.candidates[0]; // the root's children are nodes 1..k
        }
        MZ_HIP(hipMemcpyAsync(L.pool.d_start_.p, L.pool.h_start_.p, L.n * sizeof(int), hipMemcpyHostToDevice, L.stream));
        d_start = L.pool.d_start_.p;
    }
    if (!launch_select) { return MZ_OK; } // per-game simulation kernel: selection is part of the launch (runCyclesSim)
    if (resident_) { for (int g = g0; g < g1; ++g) { rotPackSet(L.rot, g - g0, games_[g].rot); } }
    int rc = L.pool.selectAsync(d_start);
    if (rc) { return rc; }
    if (resident_) {
        const double tz = nowMs();
        stats_.ms_select += tz - t0;
        return MZ_OK;
    }
    if (!(cfg_.mz_zero_copy & 2)) {
        // one D2H: [path_len n] (+ [path_action n*max_depth] for AlphaZero, whose leaves need the moves for the replay)
        MZ_HIP(hipMemcpyAsync(L.pool.h_path_arena_.p, L.pool.d_path_arena_.p,
                              (size_t(L.n) + (az ? size_t(L.n) * L.pool.v_.max_depth : 0)) * sizeof(uint32_t), hipMemcpyDeviceToHost, L.stream));
    } // else: select_kernel already wrote path_len / path_action into the pinned mirrors
    if (use_signal_) { int rcs = L.pool.signalAsync(++L.signal_seq); if (rcs) { return rcs; } }
    const double tz = nowMs();
    stats_.ms_select += tz - t0;
    trace_.add(6, tz - t0);
    return MZ_OK;
}

// phase 2 of a lane: leaf positions + feature planes on the host, then the network (asynchronously).
// device-resident cycle: leaf position + planes + legal mask, network, candidate lists, expand + backup — all queued, no host hop
int Worker::phase2Resident(Lane& L)
{
    const double t0 = nowMs();
    const int slot = sims_done_; // position slot of this simulation's leaf (slot 0 = the root)
    int rc;
    if ((rc = L.godev.leafAsync(L.pool.v_, L.rot, slot))) { return rc; }
    if ((rc = L.net->forwardAZ(reinterpret_cast<const float*>(L.godev.v_.feat), L.n, L.d_policy.p, L.d_logit.p, L.d_value.p, true))) { return rc; }
    if ((rc = L.godev.candAsync(L.pool, L.d_policy.p, L.d_logit.p, L.d_value.p, L.rot))) { return rc; }
    if ((rc = L.pool.expandBackupAsync(slot, false))) { return rc; }
    // bound the host's run-ahead (kernel-argument ring, launch queue): a completion word every 32 cycles, wait for the one before
    if (++L.cycles_since_signal >= 32) {
        L.cycles_since_signal = 0;
        if (L.signal_seq > 0 && (rc = L.pool.waitSignal(L.signal_seq))) { return rc; }
        if ((rc = L.pool.signalAsync(++L.signal_seq))) { return rc; }
    }
    stats_.ms_forward += nowMs() - t0;
    return MZ_OK;
}

int Worker::phase2(Lane& L)
{
    if (resident_) { return phase2Resident(L); }
    const bool az = desc_.type == 0;
    const int g0 = L.g0;
    const double t0 = nowMs();
    if (use_signal_) { int rcw = L.pool.waitSignal(L.signal_seq); if (rcw) { return rcw; } }
    else { MZ_HIP(hipStreamSynchronize(L.stream)); } // paths of this lane
    const double t1 = nowMs();
    stats_.ms_select += t1 - t0;
    trace_.add(7, t1 - t0);
    if (raw_bytes_ > 0 && sims_done_ == 0) { // root observations: newest screen only when every game's previous block is on the device
        bool inc = L.raw_have && L.d_raw2.p != nullptr;
        for (int j = 0; inc && j < L.n; ++j) {
            const Game& gm = games_[g0 + j];
            inc = gm.env->rawSerial() == gm.raw_seen + 1 || gm.env->rawValidCount() == 1;
        }
        L.raw_incremental = inc;
    }
    threads_->parallelFor(L.n, [this, g0](int j) { buildLeaf(g0 + j); });
    const double t2 = nowMs();
    stats_.ms_env += t2 - t1;
    trace_.add(8, t2 - t1);
    int rc;
    if (az) {
        const size_t fbytes = feat_bits_ ? size_t(L.n) * games_[0].env->featureWords() * sizeof(uint32_t) : size_t(L.n) * L.net->featSize() * sizeof(float);
        // zero-copy: the tower kernel stages its LDS tile straight from pinned host memory / the heads kernel writes the outputs there
        const bool zin = cfg_.mz_zero_copy & 1, zout = cfg_.mz_zero_copy & 2;
        if (!zin) { MZ_HIP(hipMemcpyAsync(L.d_feat.p, L.h_feat.p, fbytes, hipMemcpyHostToDevice, L.stream)); }
        if ((rc = L.net->forwardAZ(zin ? L.h_feat.p : L.d_feat.p, L.n, zout ? L.h_policy.p : L.d_policy.p, zout ? L.h_logit.p : L.d_logit.p,
                                  zout ? L.h_value.p : L.d_value.p, feat_bits_))) {
            return rc;
        }
        if (zout) {
            if (use_signal_) { if ((rc = L.pool.signalAsync(++L.signal_seq))) { return rc; } }
            const double t3z = nowMs();
            stats_.ms_forward += t3z - t2;
            trace_.add(9, t3z - t2);
            return MZ_OK;
        }
    } else if (sims_done_ == 0) {
        if (raw_bytes_ > 0 && L.raw_incremental) {
            const int fb = games_[g0].env->rawFrameBytes(), mb = raw_bytes_ - 8 * fb;
            uint8_t* prev = L.raw_cur == 0 ? L.d_raw.p : L.d_raw2.p;
            uint8_t* cur = L.raw_cur == 0 ? L.d_raw2.p : L.d_raw.p;
            MZ_HIP(hipMemcpyAsync(L.d_new.p, L.h_new.p, size_t(L.n) * fb, hipMemcpyHostToDevice, L.stream));
            MZ_HIP(hipMemcpyAsync(L.d_meta.p, L.h_meta.p, size_t(L.n) * mb, hipMemcpyHostToDevice, L.stream));
            if ((rc = L.net->shiftExpandAtariFeatures(prev, L.d_new.p, L.d_meta.p, cur, raw_bytes_, L.n, L.d_feat.p))) { return rc; }
            L.raw_cur ^= 1;
        } else if (raw_bytes_ > 0) {
            uint8_t* cur = L.raw_cur == 0 ? L.d_raw.p : L.d_raw2.p;
            MZ_HIP(hipMemcpyAsync(cur, L.h_raw.p, size_t(L.n) * raw_bytes_, hipMemcpyHostToDevice, L.stream));
            if ((rc = L.net->expandAtariFeatures(cur, raw_bytes_, L.n, L.d_feat.p))) { return rc; }
            L.raw_have = true;
        } else {
            MZ_HIP(hipMemcpyAsync(L.d_feat.p, L.h_feat.p, size_t(L.n) * L.net->featSize() * sizeof(float), hipMemcpyHostToDevice, L.stream));
        }
        if ((rc = L.pool.hiddenIndexAsync(slab_slots_, 0, L.d_src_idx.p, L.d_dst_idx.p, L.d_action_ids.p))) { return rc; }
        if ((rc = L.net->initial(L.d_feat.p, L.n, L.d_policy.p, L.d_logit.p, L.d_value.p, L.d_hidden.p, L.d_dst_idx.p))) { return rc; }
        MZ_HIP(hipMemsetAsync(L.d_reward.p, 0, L.n * sizeof(float), L.stream));
    } else {
        // device-resident MuZero step: parent hidden state gathered from the slab, action plane synthesised on device
        if ((rc = L.pool.hiddenIndexAsync(slab_slots_, sims_done_, L.d_src_idx.p, L.d_dst_idx.p, L.d_action_ids.p))) { return rc; }
        if ((rc = L.net->recurrent(L.d_hidden.p, L.d_src_idx.p, nullptr, L.d_action_ids.p, L.n, L.d_policy.p, L.d_logit.p, L.d_value.p, L.d_reward.p,
                                  L.d_hidden.p, L.d_dst_idx.p))) {
            return rc;
        }
    }
    MZ_HIP(hipMemcpyAsync(L.h_out.p, L.d_out.p, L.h_out.n * sizeof(float), hipMemcpyDeviceToHost, L.stream));
    if (use_signal_) { if ((rc = L.pool.signalAsync(++L.signal_seq))) { return rc; } }
    const double t3 = nowMs();
    stats_.ms_forward += t3 - t2;
    trace_.add(9, t3 - t2);
    return MZ_OK;
}

int Worker::cycle()
{
    const double t0 = nowMs();
    MZ_HIP(hipSetDevice(device_));
    sim_pre_ = sims_done_;
    sim_post_ = sims_done_ + 1;
    const bool root_expansion = pending_ && (sim_post_ == 1), done = pending_ && (sim_post_ == n_ + 1);
    for (auto& L : lanes_) {
        int rc = phase1(*L, root_expansion, done);
        if (rc) { return rc; }
    }
    root_host_pending_ = false;
    if (done && cfg_.mz_manual_step) { search_done_ = true; stop_now_ = true; pending_ = false; sims_done_ = 0; return MZ_OK; }
    if (pending_) { sims_done_ = done ? 0 : sim_post_; }
    for (auto& L : lanes_) {
        int rc = phase2(*L);
        if (rc) { return rc; }
    }
    pending_ = true;
    ++stats_.cycles;
    stats_.leaf_evals += G_;
    stats_.ms_total += nowMs() - t0;
    return MZ_OK;
}

// Gumbel state of the lane's games between host (Game::candidates / sample_size / simulation_budget) and device (gumbel.h layout)
int Worker::syncGumbel(Lane& L, bool to_device)
{
    const int stride = 3 + kGumbelMaxSample;
    if (to_device) {
        for (int j = 0; j < L.n; ++j) {
            const Game& gm = games_[L.g0 + j];
            int* st = L.h_gum.p + size_t(j) * stride;
            st[0] = static_cast<int>(gm.candidates.size());
            st[1] = gm.sample_size;
            st[2] = gm.simulation_budget;
            for (size_t i = 0; i < gm.candidates.size() && i < size_t(kGumbelMaxSample); ++i) { st[3 + i] = gm.candidates[i]; }
        }
        MZ_HIP(hipMemcpyAsync(L.d_gum.p, L.h_gum.p, L.h_gum.n * sizeof(int), hipMemcpyHostToDevice, L.stream));
        return MZ_OK;
    }
    MZ_HIP(hipMemcpyAsync(L.h_gum.p, L.d_gum.p, L.h_gum.n * sizeof(int), hipMemcpyDeviceToHost, L.stream));
    MZ_HIP(hipStreamSynchronize(L.stream));
    for (int j = 0; j < L.n; ++j) {
        Game& gm = games_[L.g0 + j];
        const int* st = L.h_gum.p + size_t(j) * stride;
        gm.candidates.assign(st + 3, st + 3 + std::max(0, std::min(st[0], kGumbelMaxSample)));
        gm.sample_size = st[1];
        gm.simulation_budget = st[2];
    }
    return MZ_OK;
}

// The RNG draws of the cycles [b0, b1) that join a launch: row b of the lanes' rotation tables (AlphaZero: one draw per game and cycle, zero_actor.cpp:56) and — cycle
// `noise_row` — the root noise of every game (its length is the number of legal moves, which the host engine knows; zero_actor.cpp:194-213).  Per stream the
// order is the reference's: cycle-major, actor-minor, [noise][rotation] per actor.  One stream (mz_rng_streams = 1, the deterministic contract): the calling
// thread; T streams: each is a contiguous block of games with its own generator, so the blocks are drawn side by side on the pool's threads (the 20 k gamma
// draws of 256 roots: 0.9 ms on one thread).  (Its own function: the hot loop of a 400-simulation move — 102 400 rotation draws per move on BASELINE
// configs[1] — must not depend on the inlining decisions inside runCyclesSim, which cost 1.4 ms per move once.)
__attribute__((noinline, aligned(64))) void Worker::drawStream(int t, int b0, int b1, int noise_row)
{
    const bool az = desc_.type == 0, rotate = cfg_.actor_use_random_rotation_features;
    // the games of stream t: [gs, ge) (streamOf is monotone)
    int gs = 0, ge = G_;
    if (streams_ > 1) {
        gs = static_cast<int>((static_cast<long long>(t) * G_ + streams_ - 1) / streams_);
        ge = static_cast<int>((static_cast<long long>(t + 1) * G_ + streams_ - 1) / streams_);
    }
    if (gs >= ge) { return; }
    Rng& rng = rngs_[t];
    std::vector<float>& scratch = stream_noise_[t];
    for (int b = b0; b < b1; ++b) {
        const bool noise_cycle = b == noise_row;
        if (!noise_cycle && !az) { continue; } // MuZero draws nothing in a plain cycle
        for (auto& L : lanes_) {
            const int j0 = std::max(gs, L->g0) - L->g0, j1 = std::min(ge, L->g0 + L->n) - L->g0;
            if (j0 >= j1) { continue; }
            uint8_t* rot_row = L->h_rot.p + size_t(b) * L->n;
            if (!noise_cycle) { // a plain cycle only fills its row of the rotation table (Game::rot is drawn again by the next phase1 before anybody reads it)
                for (int j = j0; j < j1; ++j) { rot_row[j] = rotate ? static_cast<uint8_t>(rng.randInt() % 8) : 0; }
                continue;
            }
            for (int j = j0; j < j1; ++j) {
                Game& gm = games_[L->g0 + j];
                gm.env->legalMask(gm.legal.data());
                int k = 0;
                for (int a = 0; a < A_; ++a) { k += gm.legal[a] != 0; }
                if (cfg_.actor_use_dirichlet_noise) { rng.dirichlet(cfg_.actor_dirichlet_noise_alpha, k, scratch); }
                else { rng.gumbel(k, scratch); }
                memcpy(L->h_noise.p + size_t(j) * A_, scratch.data(), size_t(k) * sizeof(float));
                if (az) { // AlphaZero: the only draw of a plain cycle (zero_actor.cpp:56); MuZero draws nothing
                    gm.rot = rotate ? rng.randInt() % 8 : 0;
                    rot_row[j] = static_cast<uint8_t>(gm.rot);
                }
            }
        }
    }
}

void Worker::drawCycles(int b0, int b1, int noise_row)
{
    if (b0 >= b1) { return; }
    if (streams_ == 1) { drawStream(0, b0, b1, noise_row); return; }
    threads_->parallelFor(streams_, [this, b0, b1, noise_row](int t) { drawStream(t, b0, b1, noise_row); });
}

// Device-resident cycles in batches: the host part of a cycle (per-move logic in RNG order, rotation draws) runs exactly as in
// cycle(); every following cycle that needs nothing from the host but its rotation draws joins the same launch of the per-game
// simulation kernel.  A 400-simulation move is two launches: the root expansion, then (after the root noise) the other 400.
int Worker::runCyclesSim(int n)
{
    MZ_HIP(hipSetDevice(device_));
    const bool noise_cfg = cfg_.actor_use_dirichlet_noise || cfg_.actor_use_gumbel_noise;
    int i = 0;
    while (i < n) {
        const double t0 = nowMs();
        sim_pre_ = sims_done_;
        sim_post_ = sims_done_ + 1;
        const bool root_expansion = pending_ && (sim_post_ == 1), done = pending_ && (sim_post_ == n_ + 1);
        const bool host_gumbel = dev_gumbel_ && pending_; // the host runs this cycle's Gumbel step itself: state down, step, state up
        // muzero_atari: the root's 96x96 representation is not part of the kernel; simulation 0 of a move runs as one lock-step cycle
        // (select, host planes, stand-alone kernels); its outputs are expanded on the device when simulations follow in this call (below), otherwise by
        // the next phase1 on the host; the other n simulations are one launch
        const bool root_cycle = sim_root_host_ && (!pending_ || done);
        for (auto& L : lanes_) {
            int rc = MZ_OK;
            if (host_gumbel && (rc = syncGumbel(*L, false))) { return rc; }
            if ((rc = phase1(*L, root_expansion, done, root_cycle))) { return rc; }
            if (host_gumbel && (rc = syncGumbel(*L, true))) { return rc; }
            for (int j = 0; j < L->n; ++j) { L->h_rot.p[j] = static_cast<uint8_t>(games_[L->g0 + j].rot); }
        }
        root_host_pending_ = false;
        if (done && cfg_.mz_manual_step) { search_done_ = true; pending_ = false; sims_done_ = 0; stats_.ms_total += nowMs() - t0; return i; }
        if (pending_) { sims_done_ = done ? 0 : sim_post_; }
        int sim0 = sims_done_;
        bool root_on_device = false;
        if (root_cycle) {
            for (auto& L : lanes_) {
                int rc = phase2(*L);
                if (rc) { return rc; }
            }
            { const double tf = nowMs(); flushDeferred(); trace_.add(14, nowMs() - tf); }
            pending_ = true;
            stats_.cycles += 1;
            stats_.leaf_evals += uint64_t(G_);
            i += 1;
            // mz_sim_split: when simulations follow in this call, the root is expanded on the device from the stand-alone kernels' outputs (simulation 0 of
            // the one-workgroup kernel with a given root) and the launch of simulations 1..n follows on the stream without a host round trip; the Gumbel /
            // Dirichlet noise of the root children is drawn meanwhile (its values only depend on the RNG stream and the number of legal actions).
            root_on_device = cfg_.mz_sim_split && i < n && n_ >= 1;
            if (!root_on_device) {
                root_host_pending_ = true; // the next phase1 expands the root on the host
                stats_.ms_total += nowMs() - t0;
                trace_.add(16, nowMs() - t0);
                continue;
            }
            for (auto& L : lanes_) {
                GumbelView gv = gum_;
                gv.state = L->d_gum.p;
                bool launched = false;
                int rc = L->net->simLaunchMz(L->pool, L->d_hidden.p, slab_slots_, L->d_rootfeat.p, L->d_rootlegal.p, L->d_rootturn.p, games_[0].env->numPlayers(), L->d_policy.p,
                                            L->d_logit.p, L->d_value.p, L->d_reward.p, 0, 1, &launched, noise_cfg ? L->d_noise.p : nullptr,
                                            cfg_.actor_dirichlet_noise_epsilon, cfg_.actor_use_dirichlet_noise ? 1 : 2, dev_gumbel_ ? &gv : nullptr, L->pool.d_start_.p,
                                            host_gumbel, true, 0, false, sim_mode_);
                if (rc) { return rc; }
                if (!launched) { const std::string why = mz_last_error(); setError("worker: the root expansion kernel was not launched (%s)", why.c_str()); return MZ_ERR_STATE; }
            }
            root_host_pending_ = false;
            sims_done_ = 0;
            sim0 = 1;
            trace_.add(16, nowMs() - t0);
        }
        const double tprep = nowMs();
        trace_.add(17, tprep - t0);
        // Cycles after this one join the launch while they need nothing from the host but RNG draws.  The cycle behind the root
        // expansion (sim index 1) needs the Dirichlet noise of the root children: its values only depend on the RNG stream and on
        // the NUMBER of root children = legal moves of the root position, which the host engine knows, so they are drawn here in the
        // reference's order ([noise][rotation] per actor, zero_actor.cpp:194-213 then :56) and applied by the kernel before simulation 1.
        const bool device_noise = cfg_.actor_use_dirichlet_noise || cfg_.actor_use_gumbel_noise; // the kernel applies either kind
        int batch = 1;
        while (i + batch < n && sim0 + batch < n_ + 1 && !(sim0 + batch == 1 && noise_cfg && !device_noise)) { ++batch; }
        // (root_on_device: the launch starts AT simulation 1, its noise is drawn right here)
        const bool noise_in_batch = noise_cfg && ((sim0 == 0 && batch > 1) || root_on_device);
        if (root_on_device && noise_cfg) { drawCycles(0, 1, 0); }
        // The launch goes out in up to three parts (mz_sim_split): simulation 0 needs no draw of this loop (its rotation was drawn in phase1), so it
        // runs while the host draws the root noise; a few simulations later the rest follows, whose rotation draws (AlphaZero: one per game and
        // simulation, 102 400 per move on BASELINE configs[1]) are made while the second part runs.  Same draws in the same order: nothing a record
        // could show.  The parts are queued back to back on the lane's stream; the draws of a later part travel on a second stream.
        const bool az_draws = desc_.type == 0 && cfg_.actor_use_random_rotation_features;
        int cuts[Lane::kSimParts + 1], pre_R[Lane::kSimParts] = {0}, parts = 1;
        bool pre_alt[Lane::kSimParts] = {false};
        for (int k = 0; k <= Lane::kSimParts; ++k) { cuts[k] = k == 0 ? 0 : batch; }
        if (cfg_.mz_sim_split && sim0 == 0 && batch > 1 && (noise_in_batch || az_draws)) {
            constexpr int kSecond = 16; // simulations of the middle part: 3 ms on BASELINE configs[1], three times what the draws of the rest take on this host
            cuts[parts++] = 1;
            static const long min_draws = getenv("MZ_SIM_SPLIT_MIN_DRAWS") ? atol(getenv("MZ_SIM_SPLIT_MIN_DRAWS")) : 32768; // (tests: 0 = three parts on small pools too)
            if (az_draws && batch > 1 + kSecond && long(batch - 1 - kSecond) * G_ >= min_draws) { cuts[parts++] = 1 + kSecond; }
            cuts[parts] = batch;
        }
        // Gumbel rounds (mz_sim_rounds, muzero_atari): the launch is cut at the rounds whose leaves are evaluated ahead — [evaluation of the round's leaves]
        // [its simulations, which consume them in order] — with the stretches between them as ordinary parts; all queued back to back, no host step between
        // (muzero_atari: the launch starts at simulation 1, behind the device-side root expansion; MuZero board games: at simulation 0, the root's initial inference,
        // which goes out as a part of its own in front of the first round — a call that covers the whole move)
        const bool use_rounds = !rounds_.empty() && ((root_on_device && sim0 == 1) || (!sim_root_host_ && sim_mz_ && sim0 == 0 && batch == n_ + 1 && cfg_.mz_sim_split));
        if (use_rounds) {
            parts = 0;
            int at = 0; // offset inside the batch
            for (const Round& rd : rounds_) {
                const int o = rd.s0 - sim0;
                if (o < at || o + rd.R > batch || parts + 2 > Lane::kSimParts) { continue; }
                if (o > at) { cuts[parts] = at; pre_R[parts] = 0; ++parts; }
                cuts[parts] = o; pre_R[parts] = rd.R; pre_alt[parts] = rd.alt; ++parts;
                at = o + rd.R;
            }
            if (at < batch || parts == 0) { cuts[parts] = at; pre_R[parts] = 0; ++parts; }
            cuts[parts] = batch;
            for (auto& L : lanes_) { L->pre_epoch = L->net->nextPreEpoch(); }
        }
        int drawn = 1; // rows of the rotation table (= cycles of the batch) whose draws are made
        for (int part = 0; part < parts; ++part) {
            const int c0 = cuts[part], c1 = cuts[part + 1];
            if (drawn < c1) { drawCycles(drawn, c1, (noise_in_batch && !root_on_device) ? 1 : -1); drawn = c1; }
            for (auto& L : lanes_) {
                // the first part's uploads go in front of its kernel on the lane's stream (nothing is running); later ones overlap the running part
                hipStream_t us = part == 0 ? L->stream : L->up_stream;
                bool uploaded = false;
                if (desc_.type == 0 || part == 0) { MZ_HIP(hipMemcpyAsync(L->d_rot.p + size_t(c0) * L->n, L->h_rot.p + size_t(c0) * L->n, size_t(c1 - c0) * L->n, hipMemcpyHostToDevice, us)); uploaded = true; }
                if (noise_in_batch && (root_on_device ? part == 0 : (c0 <= 1 && 1 < c1))) { MZ_HIP(hipMemcpyAsync(L->d_noise.p, L->h_noise.p, size_t(L->n) * A_ * sizeof(float), hipMemcpyHostToDevice, us)); uploaded = true; }
                if (part > 0 && uploaded) {
                    MZ_HIP(hipEventRecord(L->ev_up, us));
                    MZ_HIP(hipStreamWaitEvent(L->stream, L->ev_up, 0));
                }
                bool launched = false;
                // Gumbel rounds: ONE pair of events around all parts — they are queued back to back without a host step, and an event between two dependent
                // launches costs ~10 us of idle GPU (12 of them per muzero_atari move); otherwise one pair per part: the host's draws between two parts are not kernel time
                if (!use_rounds || part == 0) { MZ_HIP(hipEventRecord(L->ev0[use_rounds ? 0 : part], L->stream)); }
                GumbelView gv = gum_;
                gv.state = L->d_gum.p;
                const int noise_kind = cfg_.actor_use_dirichlet_noise ? 1 : 2;
                const bool hg = host_gumbel && part == 0 && !root_on_device;
                if (use_rounds) {
                    // the first round needs the noisy logits before simulation 1 runs: the noise goes out as a launch of its own
                    if (sim0 + c0 == 1 && noise_in_batch) { int rcn = L->net->simRootNoiseMz(L->n); if (rcn) { return rcn; } }
                    if (pre_R[part] > 0) {
                        bool pre = false;
                        int rcp = MZ_OK;
                        if (cfg_.mz_sim_round_batch) {
                            rcp = L->net->simPreEvalBatchMz(L->n, L->pool.v_.max_depth, sim0 + c0, pre_R[part], L->pre_epoch, &pre, cfg_.mz_sim_round_leaves, pre_alt[part]);
                            if (pre) { ++stats_.pre_batch_launches; }
                        }
                        if (!rcp && !pre) { rcp = L->net->simPreEvalMz(L->n, L->pool.v_.max_depth, sim0 + c0, pre_R[part], L->pre_epoch, &pre, pre_alt[part] || !pairsUsable(), pairsUsable()); }
                        if (rcp) { return rcp; }
                        if (pre) { ++stats_.sim_launches; ++stats_.pre_launches; }
                    }
                }
                int rc = sim_mz_ ? L->net->simLaunchMz(L->pool, L->d_hidden.p, slab_slots_, L->d_rootfeat.p, L->d_rootlegal.p, L->d_rootturn.p,
                                                      games_[0].env->numPlayers(), L->d_policy.p, L->d_logit.p, L->d_value.p, L->d_reward.p, sim0 + c0, c1 - c0,
                                                      &launched, noise_in_batch ? L->d_noise.p : nullptr, cfg_.actor_dirichlet_noise_epsilon, noise_kind,
                                                      dev_gumbel_ ? &gv : nullptr, L->pool.d_start_.p, hg, false, use_rounds ? L->pre_epoch : 0, use_rounds && noise_in_batch, sim_mode_)
                                  : L->net->simLaunch(L->pool, L->godev.v_, L->d_policy.p, L->d_logit.p, L->d_value.p, L->d_rot.p + size_t(c0) * L->n, sim0 + c0, c1 - c0,
                                                     &launched, noise_in_batch ? L->d_noise.p : nullptr, cfg_.actor_dirichlet_noise_epsilon, noise_kind,
                                                     dev_gumbel_ ? &gv : nullptr, L->pool.d_start_.p, hg);
                if (rc) { return rc; }
                if (!launched) { const std::string why = mz_last_error(); setError("worker: the simulation kernel was not launched (%s)", why.c_str()); return MZ_ERR_STATE; }
                if (!use_rounds || part == parts - 1) { MZ_HIP(hipEventRecord(L->ev1[use_rounds ? 0 : part], L->stream)); }
                if (use_rounds && part == parts - 1 && L->h_prestat.p) { int rcc = L->net->simPreCountersAsync(L->h_prestat.p); if (rcc) { return rcc; } }
                ++stats_.sim_launches;
                // the launch that completes the move's search: the root statistics the per-move host logic reads next are queued right behind it
                L->rr_ahead = false;
                static const bool rr_ahead_on = !getenv("MZ_NO_RR_AHEAD"); // (A/B switch)
                if (rr_ahead_on && part == parts - 1 && sim0 + batch == n_ + 1 && !cfg_.mz_manual_step && L->pool.stream_ == L->stream) {
                    int rcr = L->pool.rootReadLaunch();
                    if (rcr) { return rcr; }
                    L->rr_ahead = true;
                }
            }
        }
        stats_.sim_cycles += batch;
        trace_.add(15, nowMs() - tprep);
        flushDeferred(); // the record strings of the move just decided: built while the launch runs
        sims_done_ = sim0 + batch - 1;
        pending_ = true;
        stats_.cycles += batch;
        stats_.leaf_evals += uint64_t(G_) * batch;
        i += batch;
        // the batch has to finish before the next host part (root statistics) or the return; its GPU time goes to ms_forward
        float ms_gpu = 0.0f;
        const double twait = nowMs();
        for (auto& L : lanes_) {
            MZ_HIP(hipStreamSynchronize(L->stream));
            float ms = 0.0f;
            for (int part = 0; part < (use_rounds ? 1 : parts); ++part) {
                float msp = 0.0f;
                MZ_HIP(hipEventElapsedTime(&msp, L->ev0[part], L->ev1[part]));
                ms += msp;
            }
            ms_gpu = std::max(ms_gpu, ms);
        }
        stats_.ms_forward += ms_gpu;
        if (use_rounds) { adaptRounds(); }
        stats_.ms_total += nowMs() - t0;
        trace_.add(18, nowMs() - twait);
    }
    for (auto& L : lanes_) {
        int rc = L->pool.checkError();
        if (rc) { return rc; }
    }
    return n;
}

int Worker::runCycles(int n)
{
    if (!running_ || search_done_) { return 0; }
    if (sim_kernel_) { return runCyclesSim(n); }
    stop_now_ = false;
    for (int i = 0; i < n; ++i) {
        int rc = cycle();
        if (rc) { return rc; }
        if (stop_now_) { n = i; break; }
    }
    if (resident_) { // the cycles above were only queued: the call returns when they have run
        const double t0 = nowMs();
        for (auto& L : lanes_) {
            MZ_HIP(hipStreamSynchronize(L->stream));
            int rc = L->pool.checkError();
            if (rc) { return rc; }
        }
        stats_.ms_total += nowMs() - t0;
    }
    return n;
}

int Worker::popLine(char* buf, int cap)
{
    if (lines_.empty()) { return 0; }
    OutLine& front = *lines_.front();
    // Lines leave in order, complete.  A front line whose OBS tag is still being compressed is "none yet" (0), for the pop and for the size query: the caller
    // launches its next move instead of sleeping here (mz_worker_wait_lines is the call that blocks, for stop / quit / the end of a test)
    if (front.pending.load(std::memory_order_acquire) != 0) { return 0; }
    if (front.failed.load()) { // reported once: the line goes, the records behind it keep draining
        lines_.pop_front();
        setError("worker: building the OBS tag of a record failed (the record is dropped)");
        return MZ_ERR_STATE;
    }
    const std::string& s = front.text;
    const int len = static_cast<int>(s.size());
    if (!buf) { return len; } // size query: the line stays queued
    if (cap <= len) { setError("pop_line: buffer of %d bytes too small for a %d-byte line", cap, len); return MZ_ERR_CAPACITY; }
    memcpy(buf, s.data(), len);
    buf[len] = 0;
    lines_.pop_front();
    return len;
}

int Worker::waitLines() // until every queued line is complete; the number of queued lines
{
    if (obs_) { for (auto& l : lines_) { if (l->pending.load(std::memory_order_acquire) != 0) { obs_->wait(l.get()); } } }
    return static_cast<int>(lines_.size());
}

int Worker::searchAction(int g, int* action_id, int* player, int* resign) const
{
    if (g < 0 || g >= G_) { setError("search_action: game %d out of range", g); return MZ_ERR_ARG; }
    if (!search_done_) { setError("search_action: the search is not complete"); return MZ_ERR_STATE; }
    if (action_id) { *action_id = held_[g].action; }
    if (player) { *player = held_[g].player; }
    if (resign) { *resign = held_[g].resign ? 1 : 0; }
    return MZ_OK;
}

int Worker::actGame(int g, int action_id, int player) // BaseActor::act (ref base_actor.cpp:22-30)
{
    if (g < 0 || g >= G_) { setError("act: game %d out of range", g); return MZ_ERR_ARG; }
    if (!cfg_.mz_manual_step) { setError("act: the worker plays on its own (mz_manual_step=false)"); return MZ_ERR_STATE; }
    Game& gm = games_[g];
    if (!gm.env->act(action_id, player)) { return 0; }
    gm.action_info_history.resize(gm.env->actionIds().size());
    gm.action_info_history.back() = search_done_ ? held_[g].info : ActionInfo();
    // getActionInfo() runs AFTER env_.act() (base_actor.cpp:24-28): P and V are the completed search's, R is the reward of the move just played
    for (auto& kv : gm.action_info_history.back()) {
        if (kv.first == "R") { std::ostringstream oss; oss << gm.env->reward(); kv.second = oss.str(); }
    }
    ++stats_.moves;
    return 1;
}

int Worker::resetSearchAll() // ZeroActor::resetSearch (ref zero_actor.cpp:29-34) of every game
{
    MZ_HIP(hipSetDevice(device_));
    for (auto& L : lanes_) { MZ_HIP(hipStreamSynchronize(L->stream)); }
    sims_done_ = 0;
    pending_ = false;
    search_done_ = false;
    root_host_pending_ = false;
    for (auto& h : held_) { h = Held(); }
    return resetAllSearches();
}

// ZeroActor::think's `if (!isSearchDone()) { handleSearchDone(); }` (ref zero_actor.cpp:40-45): the search stops where it stands — at least the root has to be
// expanded — and the decision is taken from the simulations run so far (move decision, resign test, P / V strings: exactly what the complete search's last
// cycle does, with fewer visits).  Per-actor stepping only.
int Worker::finishSearch()
{
    if (!cfg_.mz_manual_step) { setError("finish_search: the worker plays on its own (mz_manual_step=false)"); return MZ_ERR_STATE; }
    if (search_done_) { return MZ_OK; }
    if (!pending_ || sims_done_ < 1) { setError("finish_search: the root has not been evaluated yet (run at least two cycles of the search first)"); return MZ_ERR_STATE; }
    MZ_HIP(hipSetDevice(device_));
    sim_pre_ = sims_done_;
    sim_post_ = sims_done_ + 1;
    const bool host_gumbel = dev_gumbel_;
    for (auto& L : lanes_) {
        int rc = MZ_OK;
        MZ_HIP(hipStreamSynchronize(L->stream));
        if (host_gumbel && (rc = syncGumbel(*L, false))) { return rc; }
        if ((rc = phase1(*L, false, true, false))) { return rc; } // done = true: root statistics, decision, held action (manual stepping returns before the next selection)
    }
    root_host_pending_ = false;
    search_done_ = true;
    pending_ = false;
    sims_done_ = 0;
    return MZ_OK;
}

int Worker::resetGameAt(int g) // ZeroActor::reset without the search part (ref zero_actor.cpp:23-27, base_actor.cpp:8-13)
{
    if (g < 0 || g >= G_) { setError("reset_game: game %d out of range", g); return MZ_ERR_ARG; }
    resetGame(games_[g], rngOf(g));
    return MZ_OK;
}

int Worker::emitGame(int g) // ThreadSharedData::outputGame (ref actor_group.cpp:24-50)
{
    if (g < 0 || g >= G_) { setError("emit_game: game %d out of range", g); return MZ_ERR_ARG; }
    flushDeferred();
    outputGame(games_[g]);
    return MZ_OK;
}

int Worker::envQuery(int g, int what, float* out) const
{
    if (g < 0 || g >= G_ || !out) { setError("env_query: bad arguments"); return MZ_ERR_ARG; }
    const GameEnv& e = *games_[g].env;
    switch (what) {
        case 0: *out = e.isTerminal() ? 1.0f : 0.0f; break;
        case 1: *out = static_cast<float>(e.turn()); break;
        case 2: *out = e.evalScore(false); break;
        case 3: *out = e.evalScore(true); break;
        case 4: *out = static_cast<float>(e.actionIds().size()); break;
        case 5: *out = e.reward(); break;
        default: setError("env_query: unknown query %d", what); return MZ_ERR_ARG;
    }
    return MZ_OK;
}

// BaseActor::act(const std::vector<std::string>&) (ref base_actor.cpp:32-40 -> Env::act(action_string_args), base_env.h:326-333 / atari.cpp:24-39)
int Worker::actString(int g, const char* const* args, int nargs)
{
    if (g < 0 || g >= G_) { setError("act: game %d out of range", g); return MZ_ERR_ARG; }
    if (nargs != 2 || !args || !args[0] || !args[1] || strlen(args[0]) != 1) { setError("act: expected {player char, action string} (ref base_env.h:328-329)"); return MZ_ERR_ARG; }
    const char c = args[0][0]; // charToPlayer (ref base_env.cpp:15-25)
    const int player = (c == 'B' || c == 'b') ? 1 : (c == 'W' || c == 'w') ? 2 : 0;
    const int action = games_[g].env->actionFromString(args[1]);
    if (player == 0 || action < 0 || action >= A_) { return 0; } // not an action of this game: act() fails like an illegal move
    return actGame(g, action, player);
}

// BaseActor::getActionInfoHistory() (ref base_actor.h:33-34): per move its (key, value) pairs.  Serialised as moves joined by '\x1e', inside a move
// key '\x1f' value '\x1f' ... (neither byte can occur in a tag: the values are numbers, "id:count" lists or hex)
int Worker::actionInfoHistory(int g, char* buf, int cap)
{
    if (g < 0 || g >= G_) { setError("action_info_history: game %d out of range", g); return MZ_ERR_ARG; }
    flushDeferred();
    std::string s;
    const auto& h = games_[g].action_info_history;
    for (size_t i = 0; i < h.size(); ++i) {
        if (i) { s += '\x1e'; }
        for (const auto& kv : h[i]) { s += kv.first; s += '\x1f'; s += kv.second; s += '\x1f'; }
    }
    const int len = static_cast<int>(s.size());
    if (!buf) { return len; }
    if (cap <= len) { setError("action_info_history: buffer of %d bytes too small for %d", cap, len); return MZ_ERR_CAPACITY; }
    memcpy(buf, s.data(), len);
    buf[len] = 0;
    return len;
}

int Worker::envFeatures(int g, int rotation, float* out, int capacity) const // Env::getFeatures(rotation) (ref base_env.h:88)
{
    if (g < 0 || g >= G_ || rotation < 0 || rotation > 7) { setError("env_features: bad arguments"); return MZ_ERR_ARG; }
    const int n = games_[g].env->featureSize();
    if (!out) { return n; }
    if (capacity < n) { setError("env_features: %d floats needed", n); return MZ_ERR_CAPACITY; }
    games_[g].env->features(rotation, out);
    return n;
}

int Worker::envLegalMask(int g, uint8_t* out, int capacity) const // Env::isLegalAction / getLegalActions for the player to move
{
    if (g < 0 || g >= G_ || !out || capacity < A_) { setError("env_legal_mask: bad arguments (%d bytes needed)", A_); return MZ_ERR_ARG; }
    games_[g].env->legalMask(out);
    return A_;
}

int Worker::envSetTurn(int g, int player) // Env::setTurn (ref base_env.h:103)
{
    if (g < 0 || g >= G_ || player < 1 || player > games_[g].env->numPlayers()) { setError("env_set_turn: bad arguments"); return MZ_ERR_ARG; }
    if (!cfg_.mz_manual_step) { setError("env_set_turn: the worker plays on its own (mz_manual_step=false)"); return MZ_ERR_STATE; }
    games_[g].env->setTurn(player);
    return MZ_OK;
}

int Worker::envResetSeed(int g, int seed) // Env::reset(seed) (ref atari.h:55; console.cpp:289): the seed comes from the caller, no draw
{
    if (g < 0 || g >= G_) { setError("env_reset_seed: game %d out of range", g); return MZ_ERR_ARG; }
    if (!cfg_.mz_manual_step) { setError("env_reset_seed: the worker plays on its own (mz_manual_step=false)"); return MZ_ERR_STATE; }
    games_[g].env->resetSeed(seed);
    games_[g].action_info_history.clear();
    return MZ_OK;
}

int Worker::envRotateAction(int g, int action_id, int rotation) const // Env::getRotateAction (ref base_env.h:99)
{
    if (g < 0 || g >= G_ || action_id < 0 || action_id >= A_ || rotation < 0 || rotation > 7) { setError("env_rotate_action: bad arguments"); return MZ_ERR_ARG; }
    const RotationTables* r = games_[g].env->rot();
    return r ? r->fwd[rotation][action_id] : action_id; // the Atari-shaped env has no rotations (ref atari.h:77-78)
}

int Worker::peekRecord(int game, char* buf, int cap, const char* const* keys, const char* const* values, int ntags)
{
    if (game < 0 || game >= G_) { setError("peek_record: game %d out of range", game); return MZ_ERR_ARG; }
    flushDeferred();
    ActionInfo extra;
    for (int i = 0; i < ntags; ++i) { extra.push_back({keys[i], values[i]}); }
    const std::string s = record(games_[game], extra);
    const int len = static_cast<int>(s.size());
    if (!buf) { return len; } // size query
    if (cap <= len) { setError("peek_record: buffer of %d bytes too small for a %d-byte record", cap, len); return MZ_ERR_CAPACITY; }
    memcpy(buf, s.data(), len);
    buf[len] = 0;
    return len;
}

// load_model with the file's content in hand (Network::loadModel of every network of this worker, ref actor_group.cpp:227-232, network.h:18-37)
int Worker::loadModel(const std::string& path, const mz_net_desc& file_desc, const float* weights, size_t count)
{
    MZ_HIP(hipSetDevice(device_));
    mz_net_desc a = file_desc, b = desc_;
    a.game_name[sizeof(a.game_name) - 1] = 0;
    memset(a.game_name, 0, sizeof(a.game_name)); // (the game's name is the trainer's label, not a shape)
    memset(b.game_name, 0, sizeof(b.game_name));
    if (memcmp(&a, &b, sizeof(a)) != 0) {
        setError("load_model %s: the file's hyper-parameters are not those of the running network (%s, %d blocks x %d channels)", path.c_str(),
                 desc_.game_name, desc_.num_blocks, desc_.num_hidden_channels);
        return MZ_ERR_ARG;
    }
    if (!shared_net_) {
        for (auto& L : lanes_) {
            MZ_HIP(hipStreamSynchronize(L->stream));
            int rc = L->net->reload(weights, count);
            if (rc) { return rc; }
        }
    }
    cfg_.nn_file_name = path;
    return MZ_OK;
}

int Worker::command(const std::string& line) // ref actor_group.cpp:200-252
{
    const std::string prefix = line.substr(0, line.find(' '));
    std::istringstream ign(cfg_.zero_actor_ignored_command);
    std::string tok;
    while (ign >> tok) { if (tok == prefix) { return MZ_OK; } }
    MZ_HIP(hipSetDevice(device_));
    if (prefix == "start") { running_ = true; }
    else if (prefix == "stop") { running_ = false; }
    else if (prefix == "reset_actors") {
        for (auto& gm : games_) { resetGame(gm, main_rng_); } // handleCommand runs on the main thread (actor_group.cpp:200-219)
        sims_done_ = 0;
        pending_ = false;
        for (auto& L : lanes_) { MZ_HIP(hipStreamSynchronize(L->stream)); }
        return resetAllSearches();
    } else if (prefix == "load_model") {
        if (line.find(' ') == std::string::npos) { setError("load_model needs a path"); return MZ_ERR_ARG; }
        const std::string path = line.substr(line.find(' ') + 1);
        if (shared_net_) { // the caller's Network::loadModel has (or will have) reloaded the weights; the actor only follows the name (EV tag)
            pending_weights_.clear();
            cfg_.nn_file_name = path;
            return MZ_OK;
        }
        if (pending_weights_.empty()) { // the reference's path: every network re-reads the file (actor_group.cpp:227-232)
            mz_net_desc nd;
            std::vector<float> file;
            if (!readWeightFile(path, &nd, &file)) { return MZ_ERR_ARG; }
            return loadModel(path, nd, file.data(), file.size());
        }
        std::vector<float> staged;
        staged.swap(pending_weights_);
        return loadModel(path, desc_, staged.data(), staged.size());
    } else if (prefix == "update_config") {
        if (line.find(' ') == std::string::npos) { setError("update_config needs a configuration string"); return MZ_ERR_ARG; }
        WorkerConfig nc = cfg_;
        if (!nc.loadFromString(line.substr(line.find(' ') + 1))) { return MZ_ERR_ARG; }
        // keys whose values were turned into device state when the worker was created (pool and slab sizes, PUCT tables, Gumbel constants,
        // kernels, engines): a live change would leave host and device disagreeing, so it is refused by name
        const char* fixed = nullptr;
#define MZ_FIXED(k) if (!fixed && !(nc.k == cfg_.k)) { fixed = #k; }
        MZ_FIXED(actor_num_simulation) MZ_FIXED(actor_mcts_puct_base) MZ_FIXED(actor_mcts_puct_init) MZ_FIXED(actor_mcts_reward_discount)
        MZ_FIXED(actor_mcts_value_rescale) MZ_FIXED(actor_mcts_value_flipping_player) MZ_FIXED(actor_use_gumbel) MZ_FIXED(actor_gumbel_sample_size)
        MZ_FIXED(actor_gumbel_sigma_visit_c) MZ_FIXED(actor_gumbel_sigma_scale_c) MZ_FIXED(zero_num_threads) MZ_FIXED(zero_num_parallel_games)
        MZ_FIXED(nn_type_name) MZ_FIXED(env_board_size) MZ_FIXED(env_go_komi) MZ_FIXED(env_go_ko_rule) MZ_FIXED(env_game) MZ_FIXED(atari_init_q)
        MZ_FIXED(env_atari_name) MZ_FIXED(env_atari_episode_length) MZ_FIXED(mz_pipeline_lanes) MZ_FIXED(mz_rng_streams) MZ_FIXED(mz_cpu_base) MZ_FIXED(mz_signal_wait)
        MZ_FIXED(mz_sim_kernel) MZ_FIXED(mz_sim_cluster) MZ_FIXED(mz_sim_split) MZ_FIXED(mz_sim_rounds) MZ_FIXED(mz_sim_round_min) MZ_FIXED(mz_sim_round_alt) MZ_FIXED(mz_sim_round_batch) MZ_FIXED(mz_sim_rounds_board) MZ_FIXED(mz_sim_round_pairs) MZ_FIXED(mz_sim_round_leaves) MZ_FIXED(mz_manual_step) MZ_FIXED(mz_nn_precision) MZ_FIXED(mz_raw_observations) MZ_FIXED(mz_device_env) MZ_FIXED(mz_zero_copy)
        // the Atari-shaped environments keep a window of screens sized from these three at creation (ref atari.cpp:87); records of a larger window
        // would miss frames, so they are fixed where observations are kept (board games: free to change, like the reference)
        if (games_[0].env->hasObservations()) { MZ_FIXED(zero_actor_intermediate_sequence_length) MZ_FIXED(learner_n_step_return) MZ_FIXED(learner_muzero_unrolling_step) }
#undef MZ_FIXED
        if (fixed) { setError("update_config: %s is fixed when the worker is created (restart the worker to change it)", fixed); return MZ_ERR_ARG; }
        cfg_ = nc;
    } else if (prefix == "quit") {
        running_ = false;
        return 1;
    }
    return MZ_OK; // anything else (keep_alive, ...) is silently ignored
}

} // namespace mz

// ------------------------------------------------------------------------------------------------
struct mz_worker { mz::Worker w; };
struct mz_net { mz::Net net; };
struct mz_env { std::unique_ptr<mz::GameEnv> e; };

extern "C" {

int mz_usable_cpus(void) { return mz::usableCpus(); }

mz_worker* mz_worker_create(int device, const char* conf, const mz_net_desc* desc, const float* weights, size_t count)
{
    if (!conf || (!desc != !weights)) { mz::setError("mz_worker_create: NULL argument"); return nullptr; }
    std::unique_ptr<mz_worker> w(new mz_worker());
    mz_net_desc file_desc;
    std::vector<float> file_weights;
    if (!desc) { // the network comes from the configuration's nn_file_name (ref actor_group.cpp:168-177)
        mz::WorkerConfig c;
        if (!c.loadFromString(conf)) { return nullptr; }
        if (c.nn_file_name.empty()) { mz::setError("mz_worker_create: no network given and nn_file_name is empty"); return nullptr; }
        if (!mz::readWeightFile(c.nn_file_name, &file_desc, &file_weights)) { return nullptr; }
        desc = &file_desc;
        weights = file_weights.data();
        count = file_weights.size();
    }
    if (w->w.init(device, conf, *desc, weights, count) != MZ_OK) { return nullptr; }
    if (w->w.wantsTwoLanes()) { // (decided after the first initialisation, which is what knows the kernels the shape gets; records do not depend on the lane count)
        const std::string conf2 = std::string(conf) + ":mz_pipeline_lanes=2";
        w.reset(new mz_worker());
        if (w->w.init(device, conf2.c_str(), *desc, weights, count) != MZ_OK) { return nullptr; }
    }
    return w.release();
}
mz_worker* mz_worker_create_shared(int device, const char* conf, mz_net* net)
{
    if (!conf || !net) { mz::setError("mz_worker_create_shared: NULL argument"); return nullptr; }
    std::unique_ptr<mz_worker> w(new mz_worker());
    if (w->w.init(device, conf, net->net.desc_, nullptr, 0, &net->net) != MZ_OK) { return nullptr; }
    return w.release();
}
int mz_worker_cycles_per_move(const mz_worker* w) { return w ? w->w.cyclesPerMove() : MZ_ERR_ARG; }
int mz_worker_lanes(const mz_worker* w) { return w ? w->w.numLanes() : MZ_ERR_ARG; }
void mz_worker_destroy(mz_worker* w) { delete w; }
int mz_worker_command(mz_worker* w, const char* line)
{
    if (!w || !line) { mz::setError("mz_worker_command: NULL argument"); return MZ_ERR_ARG; }
    return w->w.command(line);
}
int mz_worker_load_model(mz_worker* w, const char* path, const mz_net_desc* desc, const float* weights, size_t count)
{
    if (!w || !path || !desc || !weights) { mz::setError("mz_worker_load_model: NULL argument"); return MZ_ERR_ARG; }
    return w->w.loadModel(path, *desc, weights, count);
}
int mz_worker_set_weights(mz_worker* w, const float* weights, size_t count)
{
    if (!w || !weights) { mz::setError("mz_worker_set_weights: NULL argument"); return MZ_ERR_ARG; }
    return w->w.setWeights(weights, count);
}
int mz_worker_run_cycles(mz_worker* w, int n)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.runCycles(n);
}
int mz_worker_pop_line(mz_worker* w, char* buf, int cap)
{
    if (!w) { mz::setError("NULL argument"); return MZ_ERR_ARG; }
    return w->w.popLine(buf, cap);
}
int mz_worker_wait_lines(mz_worker* w)
{
    if (!w) { mz::setError("NULL argument"); return MZ_ERR_ARG; }
    return w->w.waitLines();
}
int mz_worker_peek_record(mz_worker* w, int game, char* buf, int cap)
{
    if (!w) { mz::setError("NULL argument"); return MZ_ERR_ARG; }
    return w->w.peekRecord(game, buf, cap);
}
int mz_worker_record(mz_worker* w, int game, const char* const* keys, const char* const* values, int ntags, char* buf, int cap)
{
    if (!w || ntags < 0 || (ntags > 0 && (!keys || !values))) { mz::setError("mz_worker_record: bad arguments"); return MZ_ERR_ARG; }
    return w->w.peekRecord(game, buf, cap, keys, values, ntags);
}
int mz_worker_search_done(const mz_worker* w) { return w ? (w->w.searchDone() ? 1 : 0) : MZ_ERR_ARG; }
int mz_worker_search_action(const mz_worker* w, int game, int* action_id, int* player, int* is_resign)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.searchAction(game, action_id, player, is_resign);
}
int mz_worker_act(mz_worker* w, int game, int action_id, int player)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.actGame(game, action_id, player);
}
int mz_worker_reset_search(mz_worker* w)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.resetSearchAll();
}
int mz_worker_finish_search(mz_worker* w)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.finishSearch();
}
int mz_worker_reset_game(mz_worker* w, int game)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.resetGameAt(game);
}
int mz_worker_emit_game(mz_worker* w, int game)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.emitGame(game);
}
int mz_worker_env_query(const mz_worker* w, int game, int what, float* out)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.envQuery(game, what, out);
}
int mz_worker_act_string(mz_worker* w, int game, const char* const* args, int nargs)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.actString(game, args, nargs);
}
int mz_worker_action_info_history(mz_worker* w, int game, char* buf, int cap)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.actionInfoHistory(game, buf, cap);
}
int mz_worker_env_features(const mz_worker* w, int game, int rotation, float* out, int capacity)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.envFeatures(game, rotation, out, capacity);
}
int mz_worker_env_legal_mask(const mz_worker* w, int game, uint8_t* out, int capacity)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.envLegalMask(game, out, capacity);
}
int mz_worker_env_set_turn(mz_worker* w, int game, int player)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.envSetTurn(game, player);
}
int mz_worker_env_reset_seed(mz_worker* w, int game, int seed)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.envResetSeed(game, seed);
}
int mz_worker_env_rotate_action(const mz_worker* w, int game, int action_id, int rotation)
{
    if (!w) { mz::setError("NULL worker"); return MZ_ERR_ARG; }
    return w->w.envRotateAction(game, action_id, rotation);
}
int mz_worker_get_stats(mz_worker* w, mz_worker_stats* out)
{
    if (!w || !out) { return MZ_ERR_ARG; }
    return w->w.getStats(out);
}
mz_net* mz_worker_net(mz_worker* w) { return w ? reinterpret_cast<mz_net*>(&w->w.net0()) : nullptr; }

mz_env* mz_env_create(const char* conf)
{
    mz::WorkerConfig c;
    if (!conf || !c.loadFromString(conf)) { return nullptr; }
    std::unique_ptr<mz_env> e(new mz_env());
    e->e = mz::createGameEnv(c.env_game, c.env_board_size, c.env_go_komi, c.env_atari_name, c.env_atari_episode_length, c.env_go_ko_rule);
    if (!e->e) { return nullptr; }
    return e.release();
}
void mz_env_destroy(mz_env* e) { delete e; }
void mz_env_reset(mz_env* e) { e->e->reset(); }
void mz_env_reset_seed(mz_env* e, int seed) { e->e->resetSeed(seed); }
float mz_env_reward(const mz_env* e) { return e->e->reward(); }
int mz_env_act(mz_env* e, int action_id, int player) { return e->e->act(action_id, player) ? 1 : 0; }
int mz_env_turn(const mz_env* e) { return e->e->turn(); }
int mz_env_is_terminal(const mz_env* e) { return e->e->isTerminal() ? 1 : 0; }
float mz_env_eval_score(const mz_env* e, int is_resign) { return e->e->evalScore(is_resign != 0); }
int mz_env_policy_size(const mz_env* e) { return e->e->policySize(); }
int mz_env_feature_size(const mz_env* e) { return e->e->featureSize(); }
int mz_env_legal_mask(const mz_env* e, uint8_t* out)
{
    e->e->legalMask(out);
    return MZ_OK;
}
int mz_env_features(const mz_env* e, int rotation, float* out)
{
    if (rotation < 0 || rotation > 7) { mz::setError("rotation %d out of range", rotation); return MZ_ERR_ARG; }
    e->e->features(rotation, out);
    return MZ_OK;
}

int mz_env_action_from_string(const mz_env* e, const char* action_string)
{
    if (!e || !action_string) { mz::setError("mz_env_action_from_string: NULL argument"); return MZ_ERR_ARG; }
    return e->e->actionFromString(action_string);
}

int mz_env_feature_bits(const mz_env* e, int rotation, uint32_t* out)
{
    if (rotation < 0 || rotation > 7) { mz::setError("rotation %d out of range", rotation); return MZ_ERR_ARG; }
    e->e->featureBits(rotation, out);
    return MZ_OK;
}

int mz_sort_candidates(int device, const float* policy, int n, int* order_out) { return mz::sortCandidatesOnDevice(device, policy, n, order_out); }

int mz_invert_values_device(int device, const float* values, int n, float* out)
{
    if (mz_device_count() < 1) { mz::setError("mz_invert_values_device: no GPU (libmzgpu has no CPU path)"); return MZ_ERR_DEVICE; }
    if (!values || !out || n < 0) { mz::setError("mz_invert_values_device: bad arguments"); return MZ_ERR_ARG; }
    return mz::invertValuesOnDevice(device, values, n, out);
}

int mz_envdev_playout(int device, const char* game, int board_size, float komi, const int* actions, int count, int root_prefix, const int* rots,
                      uint32_t* feat_out, uint8_t* legal_out, int* terminal_out, float* eval_out, int* player_out)
{
    using namespace mz;
    if (!game) { setError("mz_envdev_playout: NULL game"); return MZ_ERR_ARG; }
    if (mz_device_count() < 1) { setError("mz_envdev_playout: no GPU (libmzgpu has no CPU path)"); return MZ_ERR_DEVICE; }
    if (!actions || !rots || count < 0 || root_prefix < 0 || root_prefix > count) { setError("mz_envdev_playout: bad arguments"); return MZ_ERR_ARG; }
    const bool situational = std::string(game) == "go_situational"; // test access to env_go_ko_rule=situational
    std::unique_ptr<GameEnv> env = createGameEnv(situational ? "go" : game, board_size, komi, "ms_pacman", 1000, situational ? "situational" : "positional");
    if (!env || !env->hasDeviceTwin()) { setError("mz_envdev_playout: no device twin for this board"); return MZ_ERR_ARG; }
    for (int i = 0; i < root_prefix; ++i) {
        if (!env->act(actions[i], env->turn())) { setError("mz_envdev_playout: illegal root action %d at move %d", actions[i], i); return MZ_ERR_ARG; }
    }
    const int steps = count - root_prefix + 1, A = env->policySize(), md = steps + 1;
    const int* inv[8];
    const int* fwd[8];
    for (int r = 0; r < 8; ++r) { inv[r] = env->rot()->inv[r].data(); fwd[r] = env->rot()->fwd[r].data(); }
    GoDevice gd;
    int rc = gd.init(device, 1, env->boardSize(), komi, A, steps, md, nullptr, inv, fwd, env->zobristKeys(), env->deviceKind(), env->turnKey());
    if (rc) { return rc; }
    env->exportDeviceRoot(gd.hostSnap(0));
    if ((rc = gd.uploadRoots())) { return rc; }
    // a one-game "tree" that is a single chain: node d = the position after d device moves, kept in slot d
    DevBuf<int> d_i;
    if (!d_i.alloc(size_t(1) + 3 * md)) { setError("mz_envdev_playout: allocation failed"); return MZ_ERR_DEVICE; }
    std::vector<int> h(size_t(1) + 3 * md, 0);
    for (int d = 0; d < md; ++d) {
        h[1 + d] = d;                                                                  // path
        h[1 + md + d] = (d >= 1 && root_prefix + d - 1 < count) ? actions[root_prefix + d - 1] : -1; // path_action
        h[1 + 2 * md + d] = d;                                                         // hslot
    }
    PoolView pv{};
    pv.games = 1; pv.cap = md; pv.A = A; pv.max_depth = md;
    pv.path_len = d_i.p; pv.path = d_i.p + 1; pv.path_action = d_i.p + 1 + md; pv.hslot = d_i.p + 1 + 2 * md;
    const int W32 = (env->boardSize() * env->boardSize() + 31) / 32;
    for (int d = 0; d < steps; ++d) {
        h[0] = d + 1;
        MZ_HIP(hipMemcpy(d_i.p, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
        RotPack rp{};
        rotPackSet(rp, 0, rots[d] & 7);
        if ((rc = gd.leafAsync(pv, rp, d))) { return rc; }
        if ((rc = gd.readLeaf(feat_out + size_t(d) * env->numInputChannels() * W32, legal_out + size_t(d) * A, terminal_out + d, eval_out + d, player_out + d))) { return rc; }
    }
    return MZ_OK;
}

int mz_godev_playout(int device, int board_size, float komi, const int* actions, int count, int root_prefix, const int* rots, uint32_t* feat_out,
                     uint8_t* legal_out, int* terminal_out, float* eval_out, int* player_out)
{
    return mz_envdev_playout(device, "go", board_size, komi, actions, count, root_prefix, rots, feat_out, legal_out, terminal_out, eval_out, player_out);
}

} // extern "C"
