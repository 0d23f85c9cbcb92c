// TEST INFRASTRUCTURE ONLY.  The product's HOST code under ThreadSanitizer / AddressSanitizer + UBSan on a machine without a GPU
// (tests/csrc/Makefile builds host_check_tsan / host_check_asan; tests/test_sanitizers.py runs them in `pytest -m "not gpu"` and requires clean reports).
//
//   host_check worker "<conf>" <game_name> cin h w ch hh hw ac blocks actions vh dv type weight_seed chunk [chunk ...]
//       the worker's host half (worker.cpp through its C ABI) on tests/csrc/fake_device.cpp; a chunk "cN" runs N cycles, "!line" sends a protocol
//       command ("!stop", "!update_config k=v", "!reset_actors"), "wN" stages the synthetic parameters of seed N for the next "!load_model <name>".  Prints every finished `SelfPlay` line (L), every record as it stands (R) and the
//       counters (S): the test compares them with the oracle's ActorGroup loop (oracle_throughput_threads = the number of RNG streams).
//   host_check pool <threads> <rounds>       the spin-wait thread pool: back-to-back parallelFor calls of changing sizes, per-stream sinks merged in order
//   host_check obs <helpers> <lines>         the background compressor of the OBS tags: submit / wait / pop from a queue while helpers run, then teardown with jobs queued
//   host_check fuzz-pt <file> <iters> <seed>     byte mutations of a TorchScript archive through mz::readTorchScript
//   host_check fuzz-loader "<conf>" <file-of-records> <iters> <seed>   byte mutations of records through mz_loader_add_record
//   host_check fuzz-config <iters> <seed>    byte mutations of configuration strings through WorkerConfig::loadFromString
//   host_check fuzz-gz <iters> <seed>        compressToHex of random buffers (sizes 0 .. 200 KB)
//   host_check fuzz-env <game> <size> <iters> <seed>   random legal / illegal actions, string actions and resets through the host rules engines
#include "../../include/mzgpu.h"
#include "../../minizero_amd/csrc/config.h"
#include "../../minizero_amd/csrc/env.h"
#include "../../minizero_amd/csrc/host_threads.h"
#include "../../minizero_amd/csrc/net.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

extern "C" {
long mzo_net_param_count(const mz_net_desc* d);
void mzo_net_generate(const mz_net_desc* d, unsigned long long seed, float* out);
}

static int workerMain(int argc, char** argv)
{
    if (argc < 17) { fprintf(stderr, "usage: see the header comment\n"); return 2; }
    const std::string conf = argv[1];
    mz_net_desc d;
    memset(&d, 0, sizeof(d));
    snprintf(d.game_name, sizeof(d.game_name), "%s", argv[2]);
    int* fields[] = {&d.num_input_channels, &d.input_channel_height, &d.input_channel_width, &d.num_hidden_channels, &d.hidden_channel_height, &d.hidden_channel_width,
                     &d.num_action_feature_channels, &d.num_blocks, &d.action_size, &d.num_value_hidden_channels, &d.discrete_value_size, &d.type};
    for (int i = 0; i < 12; ++i) { *fields[i] = atoi(argv[3 + i]); }
    const unsigned long long seed = strtoull(argv[15], nullptr, 10);
    std::vector<float> w(static_cast<size_t>(mzo_net_param_count(&d)));
    mzo_net_generate(&d, seed, w.data());
    mz_worker* wk = mz_worker_create(0, conf.c_str(), &d, w.data(), w.size());
    if (!wk) { fprintf(stderr, "mz_worker_create: %s\n", mz_last_error()); return 1; }
    if (mz_worker_command(wk, "start") < 0) { fprintf(stderr, "%s\n", mz_last_error()); return 1; }
    for (int i = 16; i < argc; ++i) {
        if (argv[i][0] == 'w') { // stage the parameters of the next load_model (mz_worker_set_weights: the in-memory callers' way)
            std::vector<float> w2(w.size());
            mzo_net_generate(&d, strtoull(argv[i] + 1, nullptr, 10), w2.data());
            if (mz_worker_set_weights(wk, w2.data(), w2.size()) < 0) { fprintf(stderr, "%s\n", mz_last_error()); return 1; }
            continue;
        }
        if (argv[i][0] == '!') {
            if (mz_worker_command(wk, argv[i] + 1) < 0) { fprintf(stderr, "%s: %s\n", argv[i] + 1, mz_last_error()); return 1; }
            continue;
        }
        const int n = atoi(argv[i] + (argv[i][0] == 'c' ? 1 : 0));
        const int rc = mz_worker_run_cycles(wk, n);
        if (rc < 0) { fprintf(stderr, "run_cycles: %s\n", mz_last_error()); return 1; }
    }
    if (mz_worker_wait_lines(wk) < 0) { fprintf(stderr, "%s\n", mz_last_error()); return 1; }
    std::vector<char> buf(1 << 20);
    for (;;) {
        const int need = mz_worker_pop_line(wk, nullptr, 0);
        if (need == 0) { break; }
        if (need < 0) { fprintf(stderr, "pop_line: %s\n", mz_last_error()); return 1; }
        if (static_cast<size_t>(need) >= buf.size()) { buf.resize(static_cast<size_t>(need) + 1); }
        if (mz_worker_pop_line(wk, buf.data(), static_cast<int>(buf.size())) < 0) { fprintf(stderr, "pop_line: %s\n", mz_last_error()); return 1; }
        printf("L %s\n", buf.data());
    }
    mz_worker_stats st;
    if (mz_worker_get_stats(wk, &st) != MZ_OK) { fprintf(stderr, "%s\n", mz_last_error()); return 1; }
    const char* g = strstr(conf.c_str(), "zero_num_parallel_games=");
    const int games = g ? atoi(g + strlen("zero_num_parallel_games=")) : 0;
    for (int i = 0; i < games; ++i) {
        const int need = mz_worker_peek_record(wk, i, nullptr, 0);
        if (need < 0) { fprintf(stderr, "peek_record: %s\n", mz_last_error()); return 1; }
        if (static_cast<size_t>(need) >= buf.size()) { buf.resize(static_cast<size_t>(need) + 1); }
        if (mz_worker_peek_record(wk, i, buf.data(), static_cast<int>(buf.size())) < 0) { fprintf(stderr, "peek_record: %s\n", mz_last_error()); return 1; }
        printf("R %s\n", buf.data());
    }
    printf("S cycles=%llu leaf_evals=%llu games=%llu sim_launches=%llu\n", (unsigned long long)st.cycles, (unsigned long long)st.leaf_evals, (unsigned long long)st.games,
           (unsigned long long)st.sim_launches);
    mz_worker_destroy(wk);
    return 0;
}

// ---- thread pool: the worker's two patterns (an item per game; an item per RNG stream with a sink each, merged in order afterwards) ----
static int poolMain(int threads, int rounds)
{
    mz::ThreadPool pool(threads);
    std::mt19937 rng(12345);
    std::vector<long> out;
    long checksum = 0, expect = 0;
    for (int r = 0; r < rounds; ++r) {
        const int count = 1 + static_cast<int>(rng() % 300);
        out.assign(count, 0);
        const long salt = static_cast<long>(rng() % 1000);
        pool.parallelFor(count, [&out, salt](int i) { out[i] = salt + 3L * i; }); // (the function object dies with this statement: the epoch must be closed by then)
        for (int i = 0; i < count; ++i) { checksum += out[i]; expect += salt + 3L * i; }
        if (r % 7 == 0) { // streams with sinks
            const int S = 1 + static_cast<int>(rng() % 16);
            std::vector<std::vector<int>> sinks(S);
            pool.parallelFor(S, [&sinks, count](int t) { for (int g = t * count / static_cast<int>(sinks.size()); g < (t + 1) * count / static_cast<int>(sinks.size()); ++g) { sinks[t].push_back(g); } });
            int next = 0;
            for (auto& sk : sinks) { for (int g : sk) { if (g != next++) { fprintf(stderr, "pool: sinks out of order\n"); return 1; } } }
            if (next != count) { fprintf(stderr, "pool: %d of %d items\n", next, count); return 1; }
        }
        if (r % 97 == 0) { std::this_thread::sleep_for(std::chrono::milliseconds(3)); } // lets the workers fall asleep on the condition variable
    }
    if (checksum != expect) { fprintf(stderr, "pool: checksum %ld != %ld\n", checksum, expect); return 1; }
    printf("pool ok: %d threads, %d rounds\n", threads, rounds);
    return 0;
}

// ---- OBS compressor: lines queued in order, completed out of order by the helpers, popped in order ----
static int obsMain(int helpers, int nlines)
{
    std::deque<std::unique_ptr<mz::OutLine>> queue;
    std::mt19937 rng(99);
    static const char ph[] = "\x01OBS\x01";
    long popped = 0;
    {
        mz::ObsCompressor oc(helpers);
        for (int i = 0; i < nlines; ++i) {
            auto line = std::make_unique<mz::OutLine>();
            line->text = "SelfPlay true 1 1 0 (;GM[x]OBS[" + std::string(ph) + "]) #";
            std::string raw(1000 + rng() % 60000, '\0');
            for (auto& c : raw) { c = static_cast<char>(rng() % 7); }
            oc.submit(line.get(), std::move(raw), ph, sizeof(ph) - 1);
            queue.push_back(std::move(line));
            while (!queue.empty() && queue.front()->pending.load(std::memory_order_acquire) == 0) { // mz_worker_pop_line: complete lines leave in order
                if (queue.front()->failed.load() || queue.front()->text.find(ph) != std::string::npos) { fprintf(stderr, "obs: a line left incomplete\n"); return 1; }
                queue.pop_front();
                ++popped;
            }
        }
        for (int i = 0; i < 3 && !queue.empty(); ++i) { oc.wait(queue.front().get()); queue.pop_front(); ++popped; } // mz_worker_wait_lines
        // the compressor goes away with jobs still queued (a worker destroyed mid-run): the helpers must end before the lines do
    }
    printf("obs ok: %d helpers, %ld of %d lines popped before the teardown\n", helpers, popped, nlines);
    return 0;
}

static std::vector<uint8_t> mutate(const std::vector<uint8_t>& good, std::mt19937& rng)
{
    std::vector<uint8_t> b = good;
    const int kind = static_cast<int>(rng() % 5);
    if (kind == 0 && b.size() > 4) { b.resize(rng() % b.size()); }
    else if (kind == 1 && !b.empty()) { const size_t at = rng() % b.size(), n = 1 + rng() % 64; b.insert(b.begin() + at, n, static_cast<uint8_t>(rng())); }
    else if (kind == 2 && b.size() > 8) { const size_t at = rng() % (b.size() - 4), n = 1 + rng() % std::min<size_t>(64, b.size() - at - 1); b.erase(b.begin() + at, b.begin() + at + n); }
    else if (!b.empty()) { for (int k = 0, n = 1 + static_cast<int>(rng() % 8); k < n; ++k) { b[rng() % b.size()] = static_cast<uint8_t>((rng() % 3 == 0) ? 0xFF : ((rng() % 2) ? 0 : rng())); } }
    return b;
}

static bool slurp(const char* path, std::vector<uint8_t>* out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return false; }
    out->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

static int fuzzPt(const char* path, int iters, unsigned seed)
{
    std::vector<uint8_t> good;
    if (!slurp(path, &good)) { return 2; }
    std::mt19937 rng(seed);
    const std::string tmp = std::string(path) + ".mut";
    int ok = 0, bad = 0;
    for (int i = 0; i < iters; ++i) {
        const std::vector<uint8_t> b = i == 0 ? good : mutate(good, rng);
        { std::ofstream o(tmp, std::ios::binary); o.write(reinterpret_cast<const char*>(b.data()), static_cast<std::streamsize>(b.size())); }
        mz_net_desc d;
        std::vector<float> w;
        std::string err;
        if (mz::readTorchScript(tmp, &d, &w, &err)) { ++ok; } else { ++bad; }
        if (i == 0 && ok != 1) { fprintf(stderr, "fuzz-pt: the unmodified file does not load: %s\n", err.c_str()); return 1; }
    }
    remove(tmp.c_str());
    printf("fuzz-pt ok: %d loaded, %d refused\n", ok, bad);
    return 0;
}

static int fuzzLoader(const char* conf, const char* path, int iters, unsigned seed)
{
    std::ifstream f(path);
    std::vector<std::string> records;
    for (std::string l; std::getline(f, l);) { if (!l.empty()) { records.push_back(l); } }
    if (records.empty()) { fprintf(stderr, "fuzz-loader: no records in %s\n", path); return 2; }
    mz_loader* L = mz_loader_create(0, conf);
    if (!L) { fprintf(stderr, "mz_loader_create: %s\n", mz_last_error()); return 1; }
    std::mt19937 rng(seed);
    int loaded = 0, skipped = 0, errors = 0;
    int good = 0;
    for (const std::string& r : records) { // (a game resigned before its first move has no position to learn from: skipped, like data_loader.cpp:120-127)
        const int rc = mz_loader_add_record(L, r.c_str());
        if (rc < 0) { fprintf(stderr, "fuzz-loader: an unmodified record is an error: %s\n", mz_last_error()); mz_loader_destroy(L); return 1; }
        good += rc;
    }
    if (good == 0) { fprintf(stderr, "fuzz-loader: none of the unmodified records loads: %s\n", mz_last_error()); mz_loader_destroy(L); return 1; }
    for (int i = 0; i < iters; ++i) {
        const std::string& r = records[rng() % records.size()];
        std::vector<uint8_t> b = mutate(std::vector<uint8_t>(r.begin(), r.end()), rng);
        for (auto& c : b) { if (c == 0) { c = ' '; } } // (a C string: no embedded NUL)
        const std::string m(b.begin(), b.end());
        const int rc = mz_loader_add_record(L, m.c_str());
        if (rc == 1) { ++loaded; } else if (rc == 0) { ++skipped; } else { ++errors; }
    }
    printf("fuzz-loader ok: %d loaded, %d skipped, %d errors; %d positions in the buffer\n", loaded, skipped, errors, mz_loader_num_data(L));
    mz_loader_destroy(L);
    return 0;
}

static int fuzzConfig(int iters, unsigned seed)
{
    static const char* good[] = {
        "env_game=go:env_board_size=9:actor_num_simulation=400:zero_num_parallel_games=256:actor_use_dirichlet_noise=true:actor_dirichlet_noise_alpha=0.03",
        "env_game=atari:nn_type_name=muzero:actor_use_gumbel=true:actor_gumbel_sample_size=16:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:zero_actor_ignored_command=reset_actors keep_alive",
        "program_seed=7:program_auto_seed=false:nn_file_name=/a/b/weight_iter_100.pt:actor_mcts_value_flipping_player=W:mz_sim_kernel=true:mz_rng_streams=16"};
    std::mt19937 rng(seed);
    int ok = 0, bad = 0;
    for (int i = 0; i < iters; ++i) {
        const std::string g = good[rng() % 3];
        std::vector<uint8_t> b = i < 3 ? std::vector<uint8_t>(g.begin(), g.end()) : mutate(std::vector<uint8_t>(g.begin(), g.end()), rng);
        mz::WorkerConfig c;
        if (c.loadFromString(std::string(b.begin(), b.end()))) { ++ok; } else { ++bad; }
        if (i < 3 && ok != i + 1) { fprintf(stderr, "fuzz-config: a good string does not load: %s\n", mz_last_error()); return 1; }
    }
    printf("fuzz-config ok: %d loaded, %d refused\n", ok, bad);
    return 0;
}

static int fuzzGz(int iters, unsigned seed)
{
    std::mt19937 rng(seed);
    size_t total = 0;
    for (int i = 0; i < iters; ++i) {
        const size_t n = i == 0 ? 0 : (rng() % 8 == 0 ? rng() % 200000 : rng() % 3000);
        std::vector<uint8_t> raw(n);
        const int alphabet = 1 + static_cast<int>(rng() % 255);
        for (auto& c : raw) { c = static_cast<uint8_t>(rng() % alphabet); }
        std::string hex;
        if (!mz::compressToHex(raw.data(), raw.size(), &hex)) { fprintf(stderr, "fuzz-gz: compressToHex failed at %zu bytes\n", n); return 1; }
        if ((n == 0) != hex.empty() || hex.size() % 2) { fprintf(stderr, "fuzz-gz: %zu bytes -> %zu hex digits\n", n, hex.size()); return 1; }
        total += hex.size();
    }
    printf("fuzz-gz ok: %zu hex digits\n", total);
    return 0;
}

static int fuzzEnv(const char* game, int size, int iters, unsigned seed)
{
    std::unique_ptr<mz::GameEnv> e = mz::createGameEnv(game, size, 7.0f, "ms_pacman", 30, "positional", 1);
    if (!e) { fprintf(stderr, "fuzz-env: %s\n", mz_last_error()); return 1; }
    std::mt19937 rng(seed);
    std::vector<uint8_t> legal(static_cast<size_t>(e->policySize()));
    std::vector<float> feat(static_cast<size_t>(e->featureSize()));
    std::vector<uint32_t> bits(static_cast<size_t>(e->featureWords()) + 8);
    long moves = 0, games = 0;
    for (int i = 0; i < iters; ++i) {
        if (e->isTerminal() || rng() % 400 == 0) { if (e->needsSeed()) { e->resetSeed(static_cast<int>(rng() % 1000)); } else { e->reset(); } ++games; }
        e->legalMask(legal.data());
        int a = static_cast<int>(rng() % e->policySize());
        if (rng() % 4) { for (int k = 0; k < e->policySize(); ++k) { const int c = (a + k) % e->policySize(); if (legal[c]) { a = c; break; } } }
        const bool was_legal = legal[a] != 0;
        const bool acted = e->act(a, e->turn());
        if (acted != was_legal) { fprintf(stderr, "fuzz-env: act(%d) = %d but the legal mask says %d\n", a, int(acted), int(was_legal)); return 1; }
        moves += acted;
        if (rng() % 16 == 0) { e->features(static_cast<int>(rng() % 8), feat.data()); if (std::string(game) != "atari") { e->featureBits(static_cast<int>(rng() % 8), bits.data()); } }
        if (rng() % 64 == 0) { char s[8]; snprintf(s, sizeof(s), "%c%d", "ABCDEFGHJKLMNOPQRSTZ@1"[rng() % 22], static_cast<int>(rng() % 30) - 3); (void)e->actionFromString(s); }
        (void)e->evalScore(rng() % 2 == 0);
    }
    printf("fuzz-env ok: %s %d: %ld moves in %ld games\n", game, size, moves, games);
    return 0;
}

int main(int argc, char** argv)
{
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "worker") { return workerMain(argc - 1, argv + 1); }
    if (cmd == "pool" && argc == 4) { return poolMain(atoi(argv[2]), atoi(argv[3])); }
    if (cmd == "obs" && argc == 4) { return obsMain(atoi(argv[2]), atoi(argv[3])); }
    if (cmd == "fuzz-pt" && argc == 5) { return fuzzPt(argv[2], atoi(argv[3]), static_cast<unsigned>(atoi(argv[4]))); }
    if (cmd == "fuzz-loader" && argc == 6) { return fuzzLoader(argv[2], argv[3], atoi(argv[4]), static_cast<unsigned>(atoi(argv[5]))); }
    if (cmd == "fuzz-config" && argc == 4) { return fuzzConfig(atoi(argv[2]), static_cast<unsigned>(atoi(argv[3]))); }
    if (cmd == "fuzz-gz" && argc == 4) { return fuzzGz(atoi(argv[2]), static_cast<unsigned>(atoi(argv[3]))); }
    if (cmd == "fuzz-env" && argc == 6) { return fuzzEnv(argv[2], atoi(argv[3]), atoi(argv[4]), static_cast<unsigned>(atoi(argv[5]))); }
    fprintf(stderr, "usage: see the header comment of tests/csrc/host_check.cpp\n");
    return 2;
}
