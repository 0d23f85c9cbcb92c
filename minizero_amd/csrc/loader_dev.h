// Device entry points of the learner-side sampler (loader_kernels.hip), called by loader.cpp.
#pragma once
#include "go_dev.h"

namespace mz {

// board games: replay sample g's first d_pos[g] moves (pv.path_action) on the device engine `gd` (slot 0 = uploaded root snapshot), write the
// f32 planes of the reached position under rotation d_rot[g] to d_out [B][channels][P]
int loaderReplayFeatures(GoDevice& gd, const PoolView& pv, int B, const int* d_pos, const uint8_t* d_rot, float* d_out, hipStream_t stream);
// Atari-shaped samples: d_meta [B][kAtariMetaBytes] = 8 device pointers to 3 x 96 x 96-byte screens (oldest first), 8 f32 action values, 8 valid flags
// -> d_out [B][32][96][96]
constexpr int kAtariMetaBytes = 128;
int loaderExpandAtari(const uint8_t* d_meta, int B, float* d_out, hipStream_t stream);

} // namespace mz
