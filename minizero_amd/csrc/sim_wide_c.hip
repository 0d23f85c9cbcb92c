// instances of the wide simulation kernel, part 2 (sim_wide.inc)
#define MZ_SIM_WIDE_PART 2
#include "sim_wide.inc"
