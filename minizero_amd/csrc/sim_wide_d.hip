// instances of the wide simulation kernel, part 3 (sim_wide.inc): Othello and TicTacToe
#define MZ_SIM_WIDE_PART 3
#include "sim_wide.inc"
