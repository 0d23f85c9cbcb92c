// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
#pragma once
#include "oracle.h"
#include <mutex>

namespace mzo {

// per-cycle batch queue (ref network/alphazero_network.h:48-61, muzero_network.h:64-95)
struct NetQueue {
    const Net* net = nullptr;
    size_t feat_size = 0;
    std::vector<float> az, init, rec_h, rec_a;
    uint64_t leaf_evals = 0;
    // slot < 0: append (single-threaded, deterministic mode); slot >= 0: write into a pre-reserved slot (throughput mode)
    int pushBack(const std::vector<float>& f, int slot = -1) { return put(az, f, slot); }
    int pushBackInitial(const std::vector<float>& f, int slot = -1) { return put(init, f, slot); }
    int pushBackRecurrent(const std::vector<float>& h, const std::vector<float>& a, int slot = -1)
    {
        put(rec_a, a, slot);
        return put(rec_h, h, slot);
    }
    void reserveSlots(int B, const NetDesc& d)
    {
        reserve_b = B;
        (void)d;
    }
    int reserve_b = 0;
    std::vector<NetOutput> run();

private:
    int put(std::vector<float>& dst, const std::vector<float>& src, int slot)
    {
        if (slot < 0) { dst.insert(dst.end(), src.begin(), src.end()); return int(dst.size() / src.size()) - 1; }
        {
            std::lock_guard<std::mutex> l(mu);
            if (dst.size() < size_t(reserve_b) * src.size()) { dst.resize(size_t(reserve_b) * src.size()); }
        }
        std::copy(src.begin(), src.end(), dst.begin() + size_t(slot) * src.size());
        return slot;
    }
    std::mutex mu;
};

class Group {
public:
    Group(const Config& cfg, const NetDesc& nd, const float* raw, size_t nraw);
    void cycle();
    // one line of the stdin protocol, applied between two cycles (ref actor_group.cpp:200-252: handleCommand runs on the main thread while
    // do_cpu_job_ is true, i.e. after a GPU phase and before the next CPU phase); `raw` = the parameters of the file load_model names
    // (the oracle reads no files).  1: quit, 0: done / ignored, -1: malformed
    int command(const std::string& line, const float* raw, size_t nraw);
    bool running_ = true; // the tests drive cycle() directly; `stop` / `start` gate mzo_group_cycles like running_ gates ActorGroup::run (:140)
    Config cfg_;
    NetDesc nd_;
    std::unique_ptr<Net> net_;
    NetQueue q_;
    Random main_rng_, slave_rng_;
    std::vector<std::unique_ptr<Random>> thread_rngs_; // oracle_throughput_threads > 1: one generator per slave thread (seed + id), static partition of the actors
    std::vector<int> thread_of_;                       // ... actor -> thread
    std::vector<std::vector<std::string>> thread_lines_; // ... the lines a thread produced in the current cycle
    static thread_local int cur_thread_;
    std::vector<std::unique_ptr<ZeroActor>> actors_;
    std::vector<NetOutput> outputs_;
    std::vector<std::string> lines_, trace_lines_;
    uint64_t cycles_ = 0, games_ = 0;
    bool trace_ = false;
    std::mutex out_mutex_;

private:
    std::pair<int, int> calculateTrainingDataRange(const ZeroActor& actor) const;
    void outputGame(ZeroActor& actor);
    void handleSearchDone(int actor_id);
    void traceBefore(int i);
    void traceAfter(int i);
};

} // namespace mzo
