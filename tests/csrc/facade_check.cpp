// Test driver of the C++ facades (include/minizero/{network,actor,actor_group}.h): built by __graft_entry__.build() into tests/_bin/,
// run on the GPU by tests/test_gpu_facade.py, which compares what it prints / writes with the ctypes path and with the oracle.
//   facade_check net <weight file> <out.bin> <batch>     createNetwork -> pushBack* -> forward / initialInference / recurrentInference
//   facade_check think <conf> <moves>                    think() under actor_mcts_think_time_limit: RECORD line
//   facade_check actor <conf> <moves> [features.bin]     createNetwork + createActor + think()/act() loop, SelfPlay lines on stdout; every member of
//                                                         BaseActor / Environment the facade declares is called at least once (HIST / ENV / LEGAL lines)
#include "minizero/actor.h"
#include "minizero/actor_group.h"
#include <cstdio>
#include <cstring>

using namespace minizero;

static float pattern(uint64_t i) // deterministic 0 / 1 planes (the test regenerates them in numpy)
{
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (z >> 40) % 10 < 3 ? 1.0f : 0.0f;
}

static void put(FILE* f, const std::vector<float>& v) { fwrite(v.data(), sizeof(float), v.size(), f); }

static int runNet(const char* file, const char* out, int B)
{
    std::shared_ptr<network::Network> net = network::createNetwork(file, 0);
    std::cerr << net->toString();
    FILE* f = fopen(out, "wb");
    if (!f) { return 2; }
    const size_t fs = size_t(net->getNumInputChannels()) * net->getInputChannelHeight() * net->getInputChannelWidth();
    if (net->getNetworkTypeName() == "alphazero") {
        auto az = std::static_pointer_cast<network::AlphaZeroNetwork>(net);
        for (int b = 0; b < B; ++b) {
            std::vector<float> x(fs);
            for (size_t i = 0; i < fs; ++i) { x[i] = pattern(b * fs + i); }
            if (az->pushBack(x) != b) { return 3; }
        }
        if (az->getBatchSize() != B) { return 4; }
        auto outs = az->forward();
        if (static_cast<int>(outs.size()) != B || az->getBatchSize() != 0) { return 5; }
        for (auto& o : outs) {
            auto a = std::static_pointer_cast<network::AlphaZeroNetworkOutput>(o);
            put(f, a->policy_); put(f, a->policy_logits_); put(f, {a->value_});
        }
    } else {
        auto mzn = std::static_pointer_cast<network::MuZeroNetwork>(net);
        for (int b = 0; b < B; ++b) {
            std::vector<float> x(fs);
            for (size_t i = 0; i < fs; ++i) { x[i] = pattern(b * fs + i); }
            if (mzn->pushBackInitialData(x) != b) { return 3; }
        }
        if (mzn->getInitialInputBatchSize() != B) { return 4; }
        auto outs = mzn->initialInference();
        if (static_cast<int>(outs.size()) != B || mzn->getInitialInputBatchSize() != 0) { return 5; }
        const size_t P = size_t(net->getHiddenChannelHeight()) * net->getHiddenChannelWidth(), as = size_t(mzn->getNumActionFeatureChannels()) * P;
        for (int b = 0; b < B; ++b) {
            auto m = std::static_pointer_cast<network::MuZeroNetworkOutput>(outs[b]);
            put(f, m->policy_); put(f, m->policy_logits_); put(f, {m->value_}); put(f, m->hidden_state_);
            // action planes of action (b % action_size): board games one plane with a single 1 (pass: all 0), Atari-style nets one plane per action
            std::vector<float> act(as, 0.0f);
            const int a = b % net->getActionSize();
            if (mzn->getNumActionFeatureChannels() == 1) { if (a < static_cast<int>(P)) { act[a] = 1.0f; } }
            else { for (size_t p = 0; p < P; ++p) { act[size_t(a % mzn->getNumActionFeatureChannels()) * P + p] = 1.0f; } }
            if (mzn->pushBackRecurrentData(m->hidden_state_, act) != b) { return 6; }
        }
        if (mzn->getRecurrentInputBatchSize() != B) { return 7; }
        auto rec = mzn->recurrentInference();
        if (static_cast<int>(rec.size()) != B) { return 8; }
        for (auto& o : rec) {
            auto m = std::static_pointer_cast<network::MuZeroNetworkOutput>(o);
            put(f, m->policy_); put(f, m->policy_logits_); put(f, {m->value_}); put(f, {m->reward_}); put(f, m->hidden_state_);
        }
    }
    fclose(f);
    return 0;
}

// ActorGroup's handleSearchDone (ref actor_group.cpp:116-134) written against the per-actor surface, one actor
// action id -> the strings BaseActor::act(const std::vector<std::string>&) takes (ref utils/sgf_loader.cpp:101-108 actionIDToBoardCoordinateString)
static std::vector<std::string> actionStrings(const Action& a, int board_size, bool atari)
{
    if (atari) { // ALE's action names without their PLAYER_A_ prefix (ref atari.cpp:9-22), mixed case on purpose
        static const char* const names[18] = {"noop", "Fire", "UP", "right", "LEFT", "down", "upright", "UPLEFT", "downright", "DOWNLEFT", "upfire", "RIGHTFIRE", "leftfire",
                                              "DOWNFIRE", "uprightfire", "UPLEFTFIRE", "downrightfire", "DOWNLEFTFIRE"};
        return {std::string(1, env::playerToChar(a.getPlayer())), names[a.getActionID()]};
    }
    std::string pos = "PASS";
    if (a.getActionID() < board_size * board_size) {
        const int x = a.getActionID() % board_size, y = a.getActionID() / board_size;
        pos = std::string(1, static_cast<char>('a' + x + ('a' + x >= 'i' ? 1 : 0))) + std::to_string(y + 1); // lower case on purpose: the parser upper-cases
    }
    return {std::string(1, env::playerToChar(a.getPlayer())), pos};
}

static int runActor(const std::string& conf, int moves, const char* feat_out)
{
    config::mzgpuConfigurationString() = conf;
    const std::string file = config::mzgpuConfValue(conf, "nn_file_name");
    std::shared_ptr<network::Network> net = network::createNetwork(file, 0);
    const int n = std::stoi(config::mzgpuConfValue(conf, "actor_num_simulation"));
    std::shared_ptr<actor::BaseActor> a = actor::createActor(uint64_t(n + 1) * net->getActionSize(), net); // ref actor_group.cpp:183
    if (a->createSearch() != nullptr) { return 20; }
    if (net.use_count() < 2) { std::cerr << "setNetwork did not keep the caller's network" << std::endl; return 21; } // shared, not re-read
    const int board = net->getInputChannelHeight();
    const bool atari = net->getNetworkTypeName() == "muzero_atari";
    std::vector<char> buf(1 << 22);
    for (int m = 0; m < moves; ++m) {
        if (m % 2 == 0) {
            a->think(false, false);
        } else { // the stepping surface: one beforeNNEvaluation / afterNNEvaluation pair per simulation (ref actor_group.cpp:81-114)
            a->resetSearch();
            int pairs = 0;
            while (!a->isSearchDone()) {
                a->beforeNNEvaluation();
                if (a->getNNEvaluationBatchIndex() != 0) { return 3; }
                a->afterNNEvaluation(nullptr);
                ++pairs;
            }
            if (pairs != n + 1) { std::cerr << "search took " << pairs << " evaluations, expected " << n + 1 << std::endl; return 4; }
        }
        if (m == 0) { std::cerr << a->getSearchInfo(); }
        if (!a->isResign()) {
            const Action sa = a->getSearchAction();
            const Environment& env = static_cast<const actor::BaseActor&>(*a).getEnvironment();
            if (!env.isLegalAction(sa) || env.isLegalAction(Action(sa.getActionID(), sa.getPlayer() == env::Player::kPlayer1 ? env::Player::kPlayer2 : env::Player::kPlayer1))) { return 9; }
            // every third move goes through act(vector<string>) (ref base_actor.cpp:32-40), the others through act(Action)
            if (m % 3 == 2 ? !a->act(actionStrings(sa, board, atari)) : !a->act(sa)) { return 5; }
            if (a->act(std::vector<std::string>{"B", "?"})) { return 10; } // not an action: refused like an illegal move, nothing recorded
        }
        if (a->isResign() || a->isEnvTerminal()) {
            if (mz_worker_emit_game(a->handle(), 0) != MZ_OK) { return 6; }
            if (mz_worker_wait_lines(a->handle()) < 0) { return 7; } // (an Atari record's OBS tag is compressed in the background: pop_line never blocks)
            while (mz_worker_pop_line(a->handle(), buf.data(), static_cast<int>(buf.size())) > 0) { std::cout << buf.data() << std::endl; }
            a->reset();
        }
    }
    std::cout << "RECORD " << a->getRecord({{"XX", "tag"}}) << std::endl;
    // getActionInfoHistory (ref base_actor.h:33-34): the test rebuilds the moves' P/V/R tags of the record above from this
    std::cout << "HIST";
    for (const auto& move : a->getActionInfoHistory()) {
        std::cout << " ;";
        for (const auto& kv : move) { std::cout << kv.first << "[" << kv.second << "]"; }
    }
    std::cout << std::endl;
    // the Environment view: turn, counts, features under a rotation, rotated actions, legal actions, setTurn
    Environment& env = a->getEnvironment();
    const utils::Rotation rot = utils::Rotation::kRotation270;
    std::cout << "ENV turn=" << static_cast<int>(env.getTurn()) << " actions=" << env.getNumActions() << " terminal=" << env.isTerminal() << " eval=" << a->getEvalScore()
              << " reward=" << env.getReward() << " rot5=" << env.getRotateAction(5, rot) << std::endl;
    std::cout << "LEGAL";
    for (const Action& la : env.getLegalActions()) { std::cout << " " << la.getActionID(); }
    std::cout << std::endl;
    if (feat_out) {
        FILE* f = fopen(feat_out, "wb");
        if (!f) { return 11; }
        put(f, env.getFeatures(rot));
        fclose(f);
    }
    std::cerr << env.toString();
    const env::Player other = atari ? env::Player::kPlayer1 /* one player */ : (env.getTurn() == env::Player::kPlayer1 ? env::Player::kPlayer2 : env::Player::kPlayer1);
    env.setTurn(other);
    if (a->getEnvironment().getTurn() != other) { return 12; }
    env.reset(7); // Environment::reset(seed): a new game without a draw from the actor's generator
    if (env.getNumActions() != 0 || !a->getActionInfoHistory().empty()) { return 13; }
    // setNetwork with the same network object again = ActorGroup's load_model path (actor_group.cpp:227-232): the actor keeps its worker
    mz_worker* before = a->handle();
    a->setNetwork(net);
    if (a->handle() != before) { return 14; }
    return 0;
}

// think() under actor_mcts_think_time_limit (ref zero_actor.cpp:36-49): every move is decided when the clock says so, from the simulations run until then
static int runThink(const std::string& conf, int moves)
{
    config::mzgpuConfigurationString() = conf;
    std::shared_ptr<network::Network> net = network::createNetwork(config::mzgpuConfValue(conf, "nn_file_name"), 0);
    const int n = std::stoi(config::mzgpuConfValue(conf, "actor_num_simulation"));
    std::shared_ptr<actor::BaseActor> a = actor::createActor(uint64_t(n + 1) * net->getActionSize(), net);
    for (int m = 0; m < moves && !a->isEnvTerminal(); ++m) {
        const Action act = a->think(false, false);
        if (!a->isSearchDone()) { return 3; }
        if (a->isResign()) { break; }
        if (!a->act(act)) { return 4; }
    }
    std::cout << "RECORD " << a->getRecord() << std::endl;
    return 0;
}

// Two actors on ONE network (the reference's normal pattern: many actors per Network, actor_group.cpp:179-187), different seeds, moves interleaved: the leaves one
// actor evaluated ahead (Gumbel rounds: entries that belong to the network) must never be taken for the other's.  Prints both records.
static int runTwoActors(const std::string& conf, int seed_a, int seed_b, int moves)
{
    const std::string file = config::mzgpuConfValue(conf, "nn_file_name");
    std::shared_ptr<network::Network> net = network::createNetwork(file, 0);
    const int n = std::stoi(config::mzgpuConfValue(conf, "actor_num_simulation"));
    std::shared_ptr<actor::BaseActor> actors[2];
    const int seeds[2] = {seed_a, seed_b};
    for (int k = 0; k < 2; ++k) {
        config::mzgpuConfigurationString() = conf + ":program_seed=" + std::to_string(seeds[k]);
        actors[k] = actor::createActor(uint64_t(n + 1) * net->getActionSize(), net);
    }
    if (net.use_count() < 3) { return 21; }
    std::vector<char> buf(1 << 22);
    for (int m = 0; m < moves; ++m) {
        for (int k = 0; k < 2; ++k) {
            auto& a = actors[k];
            a->think(false, false);
            if (!a->isResign() && !a->act(a->getSearchAction())) { return 5; }
            if (a->isResign() || a->isEnvTerminal()) {
                if (mz_worker_emit_game(a->handle(), 0) != MZ_OK) { return 6; }
                if (mz_worker_wait_lines(a->handle()) < 0) { return 7; }
                while (mz_worker_pop_line(a->handle(), buf.data(), static_cast<int>(buf.size())) > 0) { std::cout << "LINE" << k << " " << buf.data() << std::endl; }
                a->reset();
            }
        }
    }
    for (int k = 0; k < 2; ++k) {
        std::cout << "RECORD" << k << " " << actors[k]->getRecord() << std::endl;
        mz_worker_stats st{};
        if (mz_worker_get_stats(actors[k]->handle(), &st) != MZ_OK) { return 8; }
        std::cout << "STATS" << k << " pre_evals=" << st.pre_evals << " pre_hits=" << st.pre_hits << std::endl;
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 6 && !strcmp(argv[1], "two")) { return runTwoActors(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5])); }
    if (argc >= 4 && !strcmp(argv[1], "think")) { return runThink(argv[2], atoi(argv[3])); }
    if (argc >= 5 && !strcmp(argv[1], "net")) { return runNet(argv[2], argv[3], atoi(argv[4])); }
    if (argc >= 4 && !strcmp(argv[1], "actor")) { return runActor(argv[2], atoi(argv[3]), argc >= 5 ? argv[4] : nullptr); }
    std::cerr << "usage: facade_check net <file> <out> <batch> | actor <conf> <moves>" << std::endl;
    return 1;
}
