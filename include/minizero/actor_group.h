// minizero::actor::ActorGroup facade over libmzgpu's worker — the `-mode sp` loop with the reference's
// stdin/stdout protocol (ref actor/actor_group.cpp:136-252, actor_group.h:46-68): one command per stdin line
// (start | stop | load_model <path> | update_config <k=v:..> | reset_actors | quit | anything else ignored),
// one `SelfPlay <terminal> <data_len> <game_len> <return> <record> #` line per finished game on stdout,
// logs on stderr.  Header-only; link with -lmzgpu -pthread.
#pragma once
#include "network.h"
#include <deque>
#include <mutex>
#include <string>
#include <thread>

namespace minizero::actor {

class ActorGroup {
public:
    // conf: the merged configuration string (conf_file lines joined with ':' then -conf_str), incl. nn_file_name and env_game
    ActorGroup(const std::string& conf, int gpu_id = 0) : conf_(conf), gpu_id_(gpu_id) {}
    ~ActorGroup() { if (worker_) { mz_worker_destroy(worker_); } }

    void run()
    {
        initialize();
        while (true) {
            if (!handleCommand()) { return; }
            if (!running_) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }
            if (mz_worker_run_cycles(worker_, 1) < 0) { std::cerr << mz_last_error() << std::endl; exit(0); }
            flushGames();
        }
    }

protected:
    virtual void initialize()
    {
        const std::string key = "nn_file_name=";
        size_t p = conf_.rfind(key);
        std::string nn_file = (p == std::string::npos) ? "" : conf_.substr(p + key.size(), conf_.find(':', p) == std::string::npos ? std::string::npos : conf_.find(':', p) - p - key.size());
        mz_net_desc desc;
        std::vector<float> w;
        if (!network::readWeightFile(nn_file, desc, w)) { exit(0); }
        worker_ = mz_worker_create(gpu_id_, conf_.c_str(), &desc, w.data(), w.size());
        if (!worker_) { std::cerr << mz_last_error() << std::endl; exit(0); }
        io_thread_ = std::thread([this]() { handleIO(); });
        io_thread_.detach();
    }
    virtual void handleIO() // ref actor_group.cpp:189-198
    {
        std::string command;
        while (getline(std::cin, command)) {
            std::lock_guard<std::mutex> lock(mutex_);
            commands_.push_back(command);
        }
        std::lock_guard<std::mutex> lock(mutex_);
        commands_.push_back("quit"); // stdin closed == the server went away
    }
    virtual bool handleCommand() // ref actor_group.cpp:200-252
    {
        std::deque<std::string> cmds;
        {
            std::lock_guard<std::mutex> lock(mutex_);
            cmds.swap(commands_);
        }
        for (const std::string& command : cmds) {
            const std::string prefix = command.substr(0, command.find(' '));
            if (prefix == "load_model" && command.find(' ') != std::string::npos) {
                mz_net_desc desc;
                std::vector<float> w;
                if (!network::readWeightFile(command.substr(command.find(' ') + 1), desc, w)) { exit(0); }
                mz_worker_set_weights(worker_, w.data(), w.size());
            }
            std::cerr << "[command] " << command << std::endl;
            const int rc = mz_worker_command(worker_, command.c_str());
            if (rc < 0) { std::cerr << mz_last_error() << std::endl; exit(0); }
            if (rc == 1) { return false; } // quit
            if (prefix == "start") { running_ = true; }
            if (prefix == "stop") { running_ = false; }
        }
        return true;
    }
    void flushGames()
    {
        static thread_local std::vector<char> buf(1 << 22);
        int n;
        while ((n = mz_worker_pop_line(worker_, buf.data(), static_cast<int>(buf.size()))) > 0) { std::cout << buf.data() << std::endl; }
    }

    std::string conf_;
    int gpu_id_;
    mz_worker* worker_ = nullptr;
    bool running_ = false;
    std::mutex mutex_;
    std::deque<std::string> commands_;
    std::thread io_thread_;
};

} // namespace minizero::actor
