// instances of the wide simulation kernel, part 1 (sim_wide.inc)
#define MZ_SIM_WIDE_PART 1
#include "sim_wide.inc"
