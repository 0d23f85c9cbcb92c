// Host-visible description of the device-side Gumbel root logic (gumbel_body.h).
#pragma once

namespace mz {

constexpr int kGumbelMaxSample = 64;

struct GumbelView {
    int* state;          // per game [3 + kGumbelMaxSample]: number of candidates, sample size, simulation budget, candidate child indices
    int sample_size;     // actor_gumbel_sample_size
    float sigma_visit_c, sigma_scale_c;
    int budget0;         // max(1, floor(n / (log2(m) * m)))                      (gumbel_zero.cpp:101)
    int num_simulation;  // n
    int pad_;
    double log2_m;       // std::log2(m) from the host's libm: next budget = floor(n / (log2(m) * size / 2)) for the CURRENT size, which is not a
                         // power of two when m is not (12 -> 6 -> 3), evaluated on the device in the same double operations (gumbel_zero.cpp:110)
};

} // namespace mz
