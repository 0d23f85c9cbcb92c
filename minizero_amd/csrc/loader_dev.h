// Device entry points of the learner-side sampler (loader_kernels.hip), called by loader.cpp.
#pragma once
#include "go_dev.h"

namespace mz {

// board games: replay sample g's first d_pos[g] moves (pv.path_action) on the device engine `gd` (slot 0 = uploaded root snapshot), write the
// f32 planes of the reached position under rotation d_rot[g] to d_out [B][channels][P]
int loaderReplayFeatures(GoDevice& gd, const PoolView& pv, int B, const int* d_pos, const uint8_t* d_rot, float* d_out, hipStream_t stream);
// Atari-shaped samples: d_raw [B][raw_bytes] (8 screens, 8 action values, 8 valid flags) -> d_out [B][32][96][96]
int loaderExpandAtari(const uint8_t* d_raw, int raw_bytes, int B, float* d_out, hipStream_t stream);

} // namespace mz
