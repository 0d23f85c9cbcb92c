// minizero::actor::ActorGroup facade over libmzgpu's worker — the `-mode sp` loop with the reference's stdin/stdout protocol
// (ref actor/actor_group.cpp:136-252, actor_group.h:46-68): one command per stdin line (start | stop | load_model <path> |
// update_config <k=v:..> | reset_actors | quit | anything else ignored), one `SelfPlay <terminal> <data_len> <game_len> <return> <record> #`
// line per finished game on stdout, logs on stderr.  Header-only; link with -lmzgpu -pthread.
//
// Like the reference, ONE process drives every visible GPU (ref actor_group.cpp:168-187: one Network per device, actor i on device
// i % G; scripts/zero-worker.sh:159-162 hands one `-mode sp` process zero_num_parallel_games = batch x #GPUs and all CUDA_VISIBLE_DEVICES):
// G = min(mz_device_count(), zero_num_parallel_games) workers, worker g owns the games {i : i % G == g} on device g, is driven by its own
// host thread and seeds program_seed + g (the reference's slave thread g seeds program_seed + g, actor_group.cpp:66-70); stdout is written
// under one mutex (actor_group.cpp:42-49).  The main thread reads stdin; every command is applied by each device thread between two moves
// of its games (the reference applies commands between two cycles of the CPU phase, actor_group.cpp:200-219: a move boundary is one of the
// cycle boundaries it could have picked — any schedule of commands against cycles is a schedule the reference can produce — and the latency
// is at most one move of the device's games).  Each device says on stderr AFTER HOW MANY CYCLES it applied a command
// ("[mzgpu] device 1: load_model ... after 357 cycles"), so a run can be replayed exactly (tests/test_gpu_iteration.py does, against the oracle).
//
// Weight files are read ONCE per process: nn_file_name at start-up and every `load_model <path>` are opened and parsed by
// the thread that reads stdin (mz_weights_read) and the parsed blob is handed to every device's worker (mz_worker_load_model) — the reference
// lets every network re-read the file (actor_group.cpp:227-232), G reads and G TorchScript parses per iteration on a G-GPU node.
#pragma once
#include "mzgpu_config.h"
#include "network.h"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>

namespace minizero::actor {

class ActorGroup {
public:
    // the reference's constructor (mode_handler.cpp:147): configuration from minizero::config (see mzgpu_config.h), all visible GPUs
    ActorGroup() : conf_(config::mzgpuCollectConfiguration()), gpu_id_(-1) {}
    // conf: the merged configuration string (conf_file lines joined with ':' then -conf_str), incl. nn_file_name and env_game;
    // gpu_id >= 0: that device only; -1: every visible device
    explicit ActorGroup(const std::string& conf, int gpu_id = -1) : conf_(conf), gpu_id_(gpu_id) {}
    virtual ~ActorGroup()
    {
        for (auto& d : devices_) { if (d.worker) { mz_worker_destroy(d.worker); } }
    }

    void run()
    {
        initialize();
        std::vector<std::thread> threads;
        for (size_t g = 0; g < devices_.size(); ++g) { threads.emplace_back([this, g]() { deviceLoop(static_cast<int>(g)); }); }
        handleIO(); // returns on quit / end of stdin
        for (auto& t : threads) { t.join(); }
        std::cerr << "[mzgpu] weight files read: " << mz_weight_file_reads() << std::endl; // one per file, whatever the number of devices
    }

    inline int getNumDevices() const { return static_cast<int>(devices_.size()); }

protected:
    struct Device {
        int gpu = 0, games = 0;
        mz_worker* worker = nullptr;
        size_t next_command = 0;
        bool running = false;
    };

    virtual void initialize() // createNeuralNetworks + createActors (ref actor_group.cpp:149-187)
    {
        if (conf_.empty()) { std::cerr << "ActorGroup: empty configuration (set minizero::config::mzgpuConfigurationString() or build next to config/configuration.h)" << std::endl; exit(0); }
        const int visible = mz_device_count();
        if (visible < 1) { std::cerr << "ActorGroup: no GPU visible (libmzgpu has no CPU path)" << std::endl; exit(0); }
        auto number = [&](const char* key, int def) { const std::string v = config::mzgpuConfValue(conf_, key); return v.empty() ? def : std::stoi(v); };
        const int total_games = number("zero_num_parallel_games", 32), seed = number("program_seed", 0), threads = number("zero_num_threads", 4);
        // MZ_DEVICE_MAP=0,0 — TEST HOOK, never set it in production: logical device g -> physical ordinal map[g], so that the multi-device
        // logic below (game split i % G, seed + g, one host thread per device, the shared stdout mutex) runs with G > 1 on a one-GPU box
        // (tests/test_gpu_protocol.py).  Workers that share a physical GPU are each sized as if they had it to themselves.  Tokens are decimal ordinals only.
        std::vector<int> map;
        if (const char* m = getenv("MZ_DEVICE_MAP")) {
            std::istringstream iss(m);
            for (std::string tok; std::getline(iss, tok, ',');) {
                const bool digits = !tok.empty() && tok.size() <= 4 && tok.find_first_not_of("0123456789") == std::string::npos;
                const int o = digits ? std::stoi(tok) : -1;
                if (o < 0 || o >= visible) { std::cerr << "ActorGroup: MZ_DEVICE_MAP token '" << tok << "' is not a device ordinal (" << visible << " visible)" << std::endl; exit(0); }
                map.push_back(o);
            }
            if (map.empty()) { std::cerr << "ActorGroup: MZ_DEVICE_MAP is set but names no device" << std::endl; exit(0); }
            std::cerr << "[mzgpu] MZ_DEVICE_MAP=" << m << " (test hook): " << map.size() << " logical device(s)" << std::endl;
        }
        const int ndev = map.empty() ? visible : static_cast<int>(map.size());
        const int G = gpu_id_ >= 0 ? 1 : std::max(1, std::min(ndev, total_games));
        devices_.resize(G);
        for (int g = 0; g < G; ++g) {
            Device& d = devices_[g];
            d.gpu = gpu_id_ >= 0 ? gpu_id_ : (map.empty() ? g : map[g]);
            d.games = (total_games - g + G - 1) / G; // |{i < total : i % G == g}| (ref actor_group.cpp:185)
            // worker g's generators: mz_rng_streams = S of them (default 1; 0 = one per slave thread of this worker), seeded program_seed + g * S + t — over all
            // devices the ids 0 .. G * S - 1 of the reference's slave threads (actor_group.cpp:66-70), each used once
            const int per_worker = std::max(1, threads / G), streams_key = number("mz_rng_streams", 1), S = std::max(1, streams_key == 0 ? per_worker : streams_key);
            const std::string conf = conf_ + ":zero_num_parallel_games=" + std::to_string(d.games) + ":program_seed=" + std::to_string(seed + g * S) +
                                     ":zero_num_threads=" + std::to_string(per_worker);
            if (g == 0) { // one read for all devices
                const std::string file = config::mzgpuConfValue(conf_, "nn_file_name");
                if (file.empty()) { std::cerr << "ActorGroup: nn_file_name is empty" << std::endl; exit(0); }
                first_weights_ = readWeights(file);
            }
            d.worker = mz_worker_create(d.gpu, conf.c_str(), mz_weights_desc(first_weights_.get()), mz_weights_data(first_weights_.get()), mz_weights_count(first_weights_.get()));
            if (!d.worker) { std::cerr << mz_last_error() << std::endl; exit(0); }
        }
        first_weights_.reset();
        std::cerr << "[mzgpu] " << total_games << " games on " << G << " GPU(s)" << std::endl;
    }

    virtual void handleIO() // ref actor_group.cpp:189-198 (here on the calling thread; the device threads do the work)
    {
        std::string command;
        while (!quit_.load() && getline(std::cin, command)) {
            const std::string prefix = command.substr(0, command.find(' '));
            if (!isIgnored(prefix)) { logLine("[command] " + command); }
            Command c{command, nullptr};
            if (prefix == "load_model" && !isIgnored(prefix) && command.find(' ') != std::string::npos) { c.weights = readWeights(command.substr(command.find(' ') + 1)); }
            {
                std::lock_guard<std::mutex> lock(mutex_);
                commands_.push_back(std::move(c));
            }
            if (prefix == "quit" && !isIgnored(prefix)) { return; }
        }
        std::lock_guard<std::mutex> lock(mutex_);
        commands_.push_back(Command{"quit", nullptr}); // stdin closed == the server went away
    }

    // Network::loadModel's read (ref network/network.h:18-37), once per file and process; a file that does not load ends the worker like the reference's assert
    static std::shared_ptr<mz_weights> readWeights(const std::string& path)
    {
        mz_weights* w = mz_weights_read(path.c_str());
        if (!w) { std::cerr << mz_last_error() << std::endl; exit(0); }
        return std::shared_ptr<mz_weights>(w, mz_weights_free);
    }

    // one stderr line, whole: the stdin thread and the device threads all log (chained << on the unbuffered std::cerr interleave inside a line)
    void logLine(const std::string& line)
    {
        std::lock_guard<std::mutex> lock(err_mutex_);
        std::cerr << line + "\n" << std::flush;
    }

    bool isIgnored(const std::string& prefix) const // zero_actor_ignored_command (ref actor_group.cpp:204-212)
    {
        std::string ignored = config::mzgpuConfValue(conf_, "zero_actor_ignored_command");
        if (ignored.empty() && conf_.find("zero_actor_ignored_command") == std::string::npos) { ignored = "reset_actors"; }
        std::istringstream iss(ignored);
        std::string tok;
        while (iss >> tok) { if (tok == prefix) { return true; } }
        return false;
    }

    // one device: apply pending commands, then one move of every game of this device (n + 1 cycles = ONE kernel launch in the worker)
    void deviceLoop(int g)
    {
        Device& d = devices_[g];
        const int chunk = std::max(1, mz_worker_cycles_per_move(d.worker));
        while (true) {
            if (!handleCommand(d)) { quit_.store(true); return; }
            if (!d.running) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }
            if (mz_worker_run_cycles(d.worker, chunk) < 0) { std::cerr << mz_last_error() << std::endl; exit(0); }
            flushGames(d);
        }
    }

    virtual bool handleCommand(Device& d) // ref actor_group.cpp:200-252
    {
        std::deque<Command> cmds;
        {
            std::lock_guard<std::mutex> lock(mutex_);
            for (; d.next_command < commands_.size(); ++d.next_command) { cmds.push_back(commands_[d.next_command]); }
        }
        for (const Command& c : cmds) {
            const std::string& command = c.line;
            const std::string prefix = command.substr(0, command.find(' '));
            const int rc = c.weights ? mz_worker_load_model(d.worker, command.substr(command.find(' ') + 1).c_str(), mz_weights_desc(c.weights.get()),
                                                            mz_weights_data(c.weights.get()), mz_weights_count(c.weights.get()))
                                     : mz_worker_command(d.worker, command.c_str()); // the worker applies zero_actor_ignored_command itself
            if (rc < 0) { std::cerr << mz_last_error() << std::endl; exit(0); }
            if (rc == 1) { drainGames(d); return false; } // quit
            if (isIgnored(prefix)) { continue; }
            if (prefix == "load_model" || prefix == "reset_actors" || prefix == "update_config" || prefix == "start" || prefix == "stop") {
                mz_worker_stats st;
                if (mz_worker_get_stats(d.worker, &st) == MZ_OK) {
                    std::ostringstream oss;
                    oss << "[mzgpu] device " << (&d - devices_.data()) << ": " << command << " after " << st.cycles << " cycles";
                    logLine(oss.str());
                }
            }
            if (prefix == "start") { d.running = true; }
            if (prefix == "stop") { d.running = false; drainGames(d); }
        }
        return true;
    }

    // stop / quit: records whose OBS tag is still being compressed are waited for (between two moves flushGames takes what is complete and moves on)
    void drainGames(Device& d)
    {
        if (mz_worker_wait_lines(d.worker) < 0) { std::cerr << mz_last_error() << std::endl; exit(0); }
        flushGames(d);
    }

    void flushGames(Device& d)
    {
        static thread_local std::vector<char> buf(1 << 20);
        while (true) {
            // size first: an Atari record carries its observations as hex (OBS tag) and can be tens of megabytes with
            // zero_actor_intermediate_sequence_length=0; a line that does not fit must never block the lines behind it
            const int need = mz_worker_pop_line(d.worker, nullptr, 0);
            if (need == 0) { break; }
            if (need == MZ_ERR_STATE) { std::cerr << mz_last_error() << std::endl; continue; } // one record lost its OBS tag: dropped and reported, the queue keeps draining
            if (need > 0 && static_cast<size_t>(need) >= buf.size()) { buf.resize(static_cast<size_t>(need) + 1); }
            const int n = need < 0 ? need : mz_worker_pop_line(d.worker, buf.data(), static_cast<int>(buf.size()));
            if (n < 0) { std::cerr << mz_last_error() << std::endl; exit(0); }
            std::lock_guard<std::mutex> lock(out_mutex_); // ref actor_group.cpp:42-49
            std::cout << buf.data() << std::endl;
        }
    }

    std::string conf_;
    int gpu_id_;
    std::vector<Device> devices_;
    std::mutex mutex_, out_mutex_, err_mutex_;
    struct Command { std::string line; std::shared_ptr<mz_weights> weights; }; // weights: the parsed file of a load_model line
    std::deque<Command> commands_;
    std::shared_ptr<mz_weights> first_weights_;
    std::atomic<bool> quit_{false};
};

} // namespace minizero::actor
