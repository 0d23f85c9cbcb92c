// instances of the wide simulation kernel, part 2 (sim_wide.inc)
#define MZ_SIM_WIDE_PART 2
#define MZ_SPEC_WAYS 4 // four remembered paths instead of sixteen (pool_body.h): 7 KB instead of 26 KB of LDS beside a 119 KB tile
#include "sim_wide.inc"
