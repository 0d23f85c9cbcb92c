// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
#pragma once
#include "oracle.h"

namespace mzo {

// per-cycle batch queue (ref network/alphazero_network.h:48-61, muzero_network.h:64-95)
struct NetQueue {
    const Net* net = nullptr;
    size_t feat_size = 0;
    std::vector<float> az, init, rec_h, rec_a;
    uint64_t leaf_evals = 0;
    int pushBack(const std::vector<float>& f) { az.insert(az.end(), f.begin(), f.end()); return int(az.size() / feat_size) - 1; }
    int pushBackInitial(const std::vector<float>& f) { init.insert(init.end(), f.begin(), f.end()); return int(init.size() / feat_size) - 1; }
    int pushBackRecurrent(const std::vector<float>& h, const std::vector<float>& a)
    {
        rec_h.insert(rec_h.end(), h.begin(), h.end());
        rec_a.insert(rec_a.end(), a.begin(), a.end());
        return int(rec_h.size() / h.size()) - 1;
    }
    std::vector<NetOutput> run();
};

class Group {
public:
    Group(const Config& cfg, const NetDesc& nd, const float* raw, size_t nraw);
    void cycle();
    Config cfg_;
    NetDesc nd_;
    std::unique_ptr<Net> net_;
    NetQueue q_;
    Random main_rng_, slave_rng_;
    std::vector<std::unique_ptr<ZeroActor>> actors_;
    std::vector<NetOutput> outputs_;
    std::vector<std::string> lines_, trace_lines_;
    uint64_t cycles_ = 0, games_ = 0;
    bool trace_ = false;

private:
    std::pair<int, int> calculateTrainingDataRange(const ZeroActor& actor) const;
    void outputGame(ZeroActor& actor);
    void handleSearchDone(int actor_id);
    void traceBefore(int i);
    void traceAfter(int i);
};

} // namespace mzo
