"""GPU tests of the device-resident Go leaf environment (go_dev.hip) and the device candidate sort.

The device engine (one position slot per tree node: leaf = parent slot + one move, labels / liberties / superko set kept
incrementally) is played move by move next to the host engine (env.cpp, itself differential-tested against the oracle's
restatement of the reference in tests/test_env_parity.py): after every move the legal mask, the bit-packed feature planes
under a random rotation, the terminal flag, the Tromp-Taylor result and the player to move must be identical (bit-exact:
integer / bit work).  Small boards make captures, ko and positional-superko repeats frequent; 8x8 has P = 64 (the pass bit
starts a new mask word); 19x19 is the maximum size."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _play_and_compare(mz, n, komi, seed, max_moves, pass_prob, root_prefix_frac, ko="positional", oracle=None):
    """oracle=None: the device engine against the product's host engine (itself tested against the oracle on the CPU, tests/test_env_parity.py);
    oracle=<oracle_lib>: the device engine DIRECTLY against the oracle's restatement of the reference's GoEnv."""
    rng = np.random.default_rng(seed)
    conf = f"env_game=go:env_board_size={n}:env_go_komi={komi}:env_go_ko_rule={ko}"
    P = n * n
    env = mz.Env(conf)
    actions = []
    while not env.is_terminal() and len(actions) < max_moves:
        legal = np.nonzero(env.legal_mask())[0]
        board = legal[legal != P]
        if len(board) == 0 or rng.random() < pass_prob:
            a = P
        else:
            a = int(rng.choice(board))
        assert env.act(a)
        actions.append(a)
    root_prefix = int(len(actions) * root_prefix_frac)
    steps = len(actions) - root_prefix + 1
    rots = rng.integers(0, 8, steps).astype(np.int32)
    feat, legal, term, ev, pl = mz.envdev_playout("go" if ko == "positional" else "go_situational", n, komi, actions, root_prefix, rots, 18, P + 1)
    ref = mz.Env(conf) if oracle is None else oracle.OracleEnv(conf)
    W32 = (P + 31) // 32

    def ref_bits(rot):
        if oracle is None:
            return ref.feature_bits(rot, 18, P)
        planes = ref.features(rot).reshape(18, P)
        assert set(np.unique(planes)) <= {0.0, 1.0}
        out = np.zeros((18, W32), np.uint32)
        for p in range(P):
            out[:, p >> 5] |= (planes[:, p] != 0).astype(np.uint32) << np.uint32(p & 31)
        return out.reshape(-1)
    for a in actions[:root_prefix]:
        assert ref.act(a)
    captures = 0
    for d in range(steps):
        where = f"board {n} seed {seed} step {d} (root_prefix {root_prefix}) actions {actions[:root_prefix + d]}"
        assert pl[d] == ref.turn(), where
        assert bool(term[d]) == ref.is_terminal(), where
        if ref.is_terminal():
            assert ev[d] == ref.eval_score(), where
        else:
            assert np.array_equal(legal[d], ref.legal_mask()), where
        assert np.array_equal(feat[d], ref_bits(int(rots[d]))), where
        if d + 1 < steps:
            assert ref.act(actions[root_prefix + d]), where
    return len(actions), captures


@pytest.mark.parametrize("n,games,max_moves", [(5, 40, 120), (7, 12, 200), (9, 10, 170), (8, 6, 140), (13, 3, 300), (19, 2, 500), (3, 20, 40)])
def test_device_engine_matches_host_engine(mz, n, games, max_moves):
    total = 0
    for g in range(games):
        moves, _ = _play_and_compare(mz, n, 7.0 if g % 2 == 0 else 6.5, 1000 * n + g, max_moves, pass_prob=0.03 if g % 3 else 0.15,
                                     root_prefix_frac=[0.0, 0.3, 0.7][g % 3])
        total += moves
    assert total > games * 5


@pytest.mark.parametrize("n,games,max_moves,ko", [(9, 6, 170, "positional"), (5, 20, 120, "positional"), (3, 30, 60, "situational"), (9, 4, 170, "situational"),
                                                  (19, 1, 400, "positional")])
def test_device_engine_matches_the_oracle_directly(mz, oracle, n, games, max_moves, ko):
    """No product code on the reference side: legal masks, planes under random rotations, terminal flags and Tromp-Taylor results of the
    device engine against the oracle's GoEnv (ref environment/go/go.cpp:132-308,703-723), both ko rules."""
    total = 0
    for g in range(games):
        moves, _ = _play_and_compare(mz, n, 7.5 if g % 2 == 0 else 6.5, 7000 * n + g, max_moves, pass_prob=0.05 if g % 3 else 0.2,
                                     root_prefix_frac=[0.0, 0.5][g % 2], ko=ko, oracle=oracle)
        total += moves
    assert total > games * 4


@pytest.mark.parametrize("n,games,max_moves", [(3, 40, 60), (5, 20, 120), (9, 4, 170)])
def test_device_engine_situational_superko(mz, n, games, max_moves):
    """env_go_ko_rule=situational on the device (a turn key XORed into the hash on every move, ref go.cpp:45-49,141,222) against the host engine"""
    for g in range(games):
        _play_and_compare(mz, n, 7.0, 31000 * n + g, max_moves, pass_prob=0.2, root_prefix_frac=[0.0, 0.3, 0.7][g % 3], ko="situational")


def test_device_engine_long_9x9_game_to_the_move_cap(mz):
    # no passes: runs into the 2 * 81 move cap (ref go.cpp:253-254), root at the start: every position comes from the device
    moves, _ = _play_and_compare(mz, 9, 7.0, 77, 400, pass_prob=0.0, root_prefix_frac=0.0)
    assert moves >= 100


@pytest.fixture(scope="module")
def ref_sort(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("refsort") / "libref_sort.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "csrc", "ref_sort.cpp")], check=True, timeout=120)
    lib = ctypes.CDLL(so)
    lib.ref_sort.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]

    def run(policy):
        p = np.ascontiguousarray(policy, np.float32)
        out = np.zeros(len(p), np.int32)
        lib.ref_sort(p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(p), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        return out
    return run


def test_device_candidate_sort_is_libstdcxx_sort(mz, ref_sort):
    """The candidate order feeds child order -> PUCT tie-breaks and the ordered init-Q sum: with tied policies only the exact
    introsort of libstdc++ reproduces the reference (ref zero_actor.cpp:225-227)."""
    rng = np.random.default_rng(5)
    cases = 0
    for it in range(160):
        n = int(rng.integers(1, 83)) if it % 5 else int(rng.integers(83, 363))
        levels = int(rng.integers(1, 4)) if it % 2 else int(rng.integers(4, 2000))
        p = (rng.integers(0, levels, n) / levels).astype(np.float32)
        if it % 7 == 0:
            p = np.sort(p)
        if it % 9 == 0:
            p = rng.random(n).astype(np.float32)  # no ties: the rank path
        assert np.array_equal(mz.sort_candidates(p), ref_sort(p)), f"case {it}: n={n} levels={levels}"
        cases += 1
    assert cases == 160


def _play_and_compare_game(mz, game, conf, n, channels, num_actions, seed, max_moves, root_prefix_frac):
    """Random legal playout on the host engine; the device engine replays the tail move by move: planes under random rotations, legal mask,
    terminal flag, result and player to move must be identical after every move."""
    rng = np.random.default_rng(seed)
    env = mz.Env(conf)
    actions = []
    while not env.is_terminal() and len(actions) < max_moves:
        legal = np.nonzero(env.legal_mask())[0]
        a = int(rng.choice(legal))
        assert env.act(a)
        actions.append(a)
    root_prefix = int(len(actions) * root_prefix_frac)
    steps = len(actions) - root_prefix + 1
    rots = rng.integers(0, 8, steps).astype(np.int32)
    feat, legal, term, ev, pl = mz.envdev_playout(game, n, 0.0, actions, root_prefix, rots, channels, num_actions)
    ref = mz.Env(conf)
    for a in actions[:root_prefix]:
        assert ref.act(a)
    for d in range(steps):
        where = f"{game} {n} seed {seed} step {d} (root_prefix {root_prefix}) actions {actions[:root_prefix + d]}"
        assert pl[d] == ref.turn(), where
        assert bool(term[d]) == ref.is_terminal(), where
        if ref.is_terminal():
            assert ev[d] == ref.eval_score(), where
        else:
            assert np.array_equal(legal[d], ref.legal_mask()), where
        assert np.array_equal(feat[d], ref.feature_bits(int(rots[d]), channels, n * n)), where
        if d + 1 < steps:
            assert ref.act(actions[root_prefix + d]), where
    return len(actions)


@pytest.mark.parametrize("n", [8, 6, 4])
def test_othello_device_engine_matches_host_engine(mz, n):
    """othLeafBody: flips in all 8 directions, forced passes, the two-pass end, disc-count results — whole random games, several root depths."""
    total = 0
    for g in range(24):
        total += _play_and_compare_game(mz, "othello", f"env_game=othello:env_board_size={n}", n, 4, n * n + 1, 100 * n + g, 200, (g % 4) / 4.0)
    assert total > 24 * (n * n - 8) // 2


def test_tictactoe_device_engine_matches_host_engine(mz):
    total = 0
    for g in range(60):
        total += _play_and_compare_game(mz, "tictactoe", "env_game=tictactoe", 3, 4, 9, 7 + g, 9, (g % 3) / 3.0)
    assert total >= 60 * 5


def test_device_known_answers(mz):
    """tests/test_go_known_answers.py on the DEVICE engine (go_dev.hip): positions whose outcome follows from the Tromp-Taylor rule text (capture, suicide,
    ko / positional superko, area scoring read through the winner under two komi values) and the reference's empty-board quirk — every move made on the device
    (root_prefix = 0), the legal mask / terminal flag / result read after each."""
    import test_go_known_answers as K
    n, PASS, p = K.N, K.PASS, K.p

    def run(moves, komi=0.5):
        feat, legal, term, ev, pl = mz.godev_playout(n, komi, moves, 0, [0] * (len(moves) + 1))
        return legal, term, ev, pl  # index s = the position after s moves

    legal, term, ev, pl = run(K.CAPTURE + [PASS])
    k = len(K.CAPTURE)
    assert legal[k - 1][p(2, 2)] == 0                      # occupied by white before the capture
    assert pl[k] == 2 and legal[k][p(2, 2)] == 0           # white: suicide
    assert legal[k][p(0, 0)] == 1 and legal[k][PASS] == 1
    assert pl[k + 1] == 1 and legal[k + 1][p(2, 2)] == 1   # black may fill its own eye
    legal, term, ev, pl = run(K.KO + [p(4, 4), p(4, 0), p(1, 1)])
    k = len(K.KO)
    assert pl[k] == 2 and legal[k][p(1, 1)] == 0           # immediate recapture
    assert legal[k + 2][p(1, 1)] == 1                      # after a threat and its answer
    assert pl[k + 3] == 1 and legal[k + 3][p(1, 2)] == 0   # and now black may not retake at once
    legal, term, ev, pl = run([PASS, p(0, 1), PASS, p(1, 0)])
    assert legal[4][p(0, 0)] == 0                          # multi-stone suicide in the corner
    legal, term, ev, pl = run([p(0, 2), p(0, 1), p(1, 1), p(1, 0), p(2, 0), PASS, p(0, 0)])
    assert legal[6][p(0, 0)] == 1 and legal[7][p(0, 1)] == 0 and legal[7][p(1, 0)] == 0  # the capture makes the liberty; both emptied points are suicide for white

    def area_difference_is(moves, diff):
        for komi, want in ((diff - 0.5, 1.0), (diff + 0.5, -1.0), (float(diff), 0.0)):
            legal, term, ev, pl = run(moves, komi)
            assert term[len(moves)] == 1 and not term[:len(moves)].any()
            assert ev[len(moves)] == want, (moves, komi, ev[len(moves)])

    area_difference_is([PASS, PASS], 25)                   # the reference's empty-board quirk (go.cpp:713)
    area_difference_is([p(2, 2), PASS, PASS], 25)
    area_difference_is([p(2, 2), p(0, 0), PASS, PASS], 0)
    area_difference_is(K.CAPTURE + [PASS, PASS], 4)
    area_difference_is([p(0, 2), p(0, 4), p(1, 2), PASS, p(2, 2), PASS, p(3, 2), PASS, p(4, 2), PASS, PASS], 14)
    walls = [m for r in range(5) for m in (p(r, 1), p(r, 3))]
    area_difference_is(walls + [PASS, PASS], 0)
    area_difference_is(walls + [p(2, 2), PASS, PASS], 1)
