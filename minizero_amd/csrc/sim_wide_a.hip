// instances of the wide simulation kernel, part 0 (sim_wide.inc) + the host side of all parts
#define MZ_SIM_WIDE_PART 0
#include "sim_wide.inc"

namespace mz {

bool simWideLaunchPart1(int H, int W, int c0q, int C, int cpl, const SimArgs* d_args, int games, const uint8_t* d_rot, int sim0, int nsims, int host_start, int lf, size_t lds,
                        hipStream_t s, size_t* tile_bytes, int* rc, int* spec_words);
bool simWideLaunchPart2(int H, int W, int c0q, int C, int cpl, const SimArgs* d_args, int games, const uint8_t* d_rot, int sim0, int nsims, int host_start, int lf, size_t lds,
                        hipStream_t s, size_t* tile_bytes, int* rc, int* spec_words);
bool simWideLaunchPart3(int H, int W, int c0q, int C, int cpl, const SimArgs* d_args, int games, const uint8_t* d_rot, int sim0, int nsims, int host_start, int lf, size_t lds,
                        hipStream_t s, size_t* tile_bytes, int* rc, int* spec_words);

static int simWideCpl(int env_kind, int board_n) { return env_kind == 2 ? -1 : env_kind == 1 ? 0 : (board_n * board_n + 63) / 64; }

static bool simWideAny(int H, int W, int c0q, int C, int cpl, const SimArgs* d_args, int games, const uint8_t* d_rot, int sim0, int nsims, int host_start, int lf, size_t lds,
                       hipStream_t s, size_t* tile_bytes, int* rc, int* spec_words = nullptr)
{
    return simWideLaunchPart0(H, W, c0q, C, cpl, d_args, games, d_rot, sim0, nsims, host_start, lf, lds, s, tile_bytes, rc, spec_words) ||
           simWideLaunchPart1(H, W, c0q, C, cpl, d_args, games, d_rot, sim0, nsims, host_start, lf, lds, s, tile_bytes, rc, spec_words) ||
           simWideLaunchPart2(H, W, c0q, C, cpl, d_args, games, d_rot, sim0, nsims, host_start, lf, lds, s, tile_bytes, rc, spec_words) ||
           simWideLaunchPart3(H, W, c0q, C, cpl, d_args, games, d_rot, sim0, nsims, host_start, lf, lds, s, tile_bytes, rc, spec_words);
}

// The LDS plan of sim_kernel_wide for a search of n simulations on a board of board_n x board_n points: false = no instance, or the mandatory blocks do not fit.
// *lf = the optional blocks that fit, in the order of what they buy (superko table, the leaf's block beside the heads, path speculation); *lds = the bytes to ask for
// env_kind: 0 Go, 1 Othello, 2 TicTacToe (the leaf bodies of go_body.h; the kernels' words-per-plane argument is 0 / -1 for the latter two)
bool Net::simWidePlan(int board_n, int env_kind, int num_simulation, const HeadParams& hp, int channels, int W32, size_t leaf_bytes, size_t scratch_bytes, int* lf, size_t* lds,
                      size_t* tile_bytes_out) const
{
    TowerArgs ta;
    int c0q = 0;
    if (desc_.type != 0 || !makeWideArgs(repr_, true, &ta, &c0q)) { return false; }
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    if (H != board_n || W != board_n) { return false; }
    const int cpl = simWideCpl(env_kind, board_n);
    size_t tile_bytes = 0;
    int rc = MZ_OK, spec_words = kSpecWords;
    if (!simWideAny(H, W, c0q, C, cpl, nullptr, 0, nullptr, 0, 0, 0, 0, 0, nullptr, &tile_bytes, &rc, &spec_words)) { return false; }
    const size_t rcp_n = size_t(num_simulation) + 5, max_depth = size_t(num_simulation) + 3, A = desc_.action_size;
    const size_t heads = (size_t(hp.PC) * hp.P + hp.P + hp.VH + hp.A + 16) * sizeof(float);
    size_t need = tile_bytes + rcp_n * sizeof(double) + (2 * max_depth + 2 + ((simXchgWords(int(A), channels, W32) + 1) & ~size_t(1))) * sizeof(float) + heads + 16;
    const size_t cap = size_t(160) * 1024;
    if (need > cap || scratch_bytes > tile_bytes) { return false; }
    int f = 0;
    const size_t seen = size_t(kGoSeenCap) * sizeof(uint64_t);
    if (cpl > 0 && need + seen <= cap) { f |= 2; need += seen; } // (the superko table and the leaf's block are Go's)
    // (the walk's speculation before the leaf's block: it is what keeps the slowest game of a launch short — 9x9 x 256 without it: select + leaf 26 us on average,
    //  98 us in the deepest game, and a launch lasts as long as its slowest game)
    const size_t spec = rcp_n * (sizeof(double) + sizeof(float)) + size_t(spec_words) * sizeof(int) + 8;
    if (need + spec <= cap) { f |= 1; need += spec; }
    const bool beside = hp.VH <= 256 && (hp.PC + 1) * hp.P <= 384 && hp.A <= 384; // waves 6 and 7 have no share of the heads (sim_az_body.h)
    const size_t leaf = (leaf_bytes + 7) & ~size_t(7);
    if (beside && (f & 2) && need + leaf <= cap) { f |= 4; need += leaf; }
    if (lf) { *lf = f; }
    if (lds) { *lds = need; }
    if (tile_bytes_out) { *tile_bytes_out = tile_bytes; }
    return true;
}

int Net::simLaunchWide(const SimArgs& a, const GoDevView& gv, int max_depth, const uint8_t* d_rot, int sim0, int nsims, bool host_start, int lf, size_t lds, bool* launched)
{
    *launched = false;
    int rc = uploadSimArgs(a);
    if (rc) { return rc; }
    int c0q = 16 * repr_[0].cq;
    const int H = desc_.hidden_channel_height, W = desc_.hidden_channel_width, C = desc_.num_hidden_channels;
    if (simWideAny(H, W, c0q, C, simWideCpl(gv.kind, gv.n), reinterpret_cast<const SimArgs*>(sim_args_.p), gv.games, d_rot, sim0, nsims, host_start ? 1 : 0, lf, lds, stream_, nullptr, &rc)) {
        *launched = rc == MZ_OK;
    }
    return rc;
}

} // namespace mz
