"""Go known answers from OUTSIDE the repository and the reference: positions whose outcome follows from the rule text alone — the Tromp-Taylor rules
(area scoring: a point counts for a colour if it is that colour or only that colour can be reached from it through empty points; no suicide; a play
may not recreate an earlier whole-board position — positional superko) and the reference's two ending rules (two consecutive passes; more than
2 * N * N moves, ref environment/go/go.cpp:246-257).  Checked against BOTH CPU rules engines — the oracle's (block / liberty bitsets, oracle/o_env.cpp)
and the product's host engine (flat board + flood fill, minizero_amd/csrc/env.cpp) — and, with a GPU, against the device engine (go_dev.hip) in
tests/test_gpu_godev.py::test_device_known_answers.  The search oracle stays "parity unpinned" (DESIGN.md 5); this narrows what an unpinned rules
engine could hide.  Scores are read through the winner under two komi values that bracket the hand-counted area difference."""
import numpy as np
import pytest

N = 5
PASS = N * N


def p(r, c, n=N):
    return r * n + c


@pytest.fixture(params=["oracle", "product"])
def make_env(request, oracle, mz):
    def make(n=N, komi=0.5, rule="positional"):
        conf = f"env_game=go:env_board_size={n}:env_go_komi={komi}:env_go_ko_rule={rule}"
        return oracle.OracleEnv(conf) if request.param == "oracle" else mz.Env(conf)
    return make


def play(env, moves):
    for m in moves:
        assert env.act(m), f"move {m} refused"


def outcome(make_env, moves, komi):
    e = make_env(komi=komi)
    play(e, moves)
    assert e.is_terminal()
    return e.eval_score(False)


def area_difference_is(make_env, moves, diff):
    """black area - white area == diff: black wins at komi diff - 0.5, loses at diff + 0.5, and komi == diff is a draw"""
    assert outcome(make_env, moves, diff - 0.5) == 1.0
    assert outcome(make_env, moves, diff + 0.5) == -1.0
    assert outcome(make_env, moves, float(diff)) == 0.0


# ---- capture, suicide ----
CAPTURE = [p(1, 2), p(2, 2), p(2, 1), PASS, p(2, 3), p(4, 4), p(3, 2)]  # black surrounds the white stone at the centre; white's 3rd move is elsewhere


def test_capture_and_suicide(make_env):
    e = make_env()
    play(e, CAPTURE[:-1])
    assert e.turn() == 1 and e.legal_mask()[p(2, 2)] == 0  # occupied
    play(e, CAPTURE[-1:])                                   # the capture
    assert e.turn() == 2
    m = e.legal_mask()
    assert m[p(2, 2)] == 0, "white may not play into the eye: no liberties, nothing captured (suicide)"
    assert not e.act(p(2, 2))
    assert m[p(0, 0)] == 1 and m[PASS] == 1
    play(e, [PASS])
    assert e.legal_mask()[p(2, 2)] == 1, "black may fill its own eye (three liberties of the joined group remain)"


def test_capture_counts_in_the_area(make_env):
    # after the capture: black 4 stones, white 1 stone at (4,4); every empty region touches both colours except the eye (2,2): black 4 + 1, white 1
    area_difference_is(make_env, CAPTURE + [PASS, PASS], 4)


def test_multi_stone_suicide(make_env):
    # corner: white stones (0,1) (1,0) (1,1)... black plays (0,0) with no liberty and captures nothing -> illegal
    e = make_env()
    play(e, [PASS, p(0, 1), PASS, p(1, 0)])
    assert e.turn() == 1 and e.legal_mask()[p(0, 0)] == 0 and not e.act(p(0, 0))


def test_capture_makes_the_liberty(make_env):
    # the same corner point IS legal when it captures: black (0,2) (1,1) (2,0) take the liberties of white (0,1) (1,0) first
    e = make_env()
    play(e, [p(0, 2), p(0, 1), p(1, 1), p(1, 0), p(2, 0), PASS])
    assert e.legal_mask()[p(0, 0)] == 1
    play(e, [p(0, 0)])  # captures both white stones
    m = e.legal_mask()
    assert m[p(0, 1)] == 0 and m[p(1, 0)] == 0, "white: both points are suicide now"
    # black 4 stones + the two emptied points; the rest of the board touches only black too: 25 - 0
    play(e, [PASS, PASS])
    assert e.is_terminal()


# ---- ko ----
#   . B W .      black plays (1,2): captures (1,1); white's recapture at (1,1) would recreate the position before -> illegal until the board differs elsewhere
#   B W . W
#   . B W .
KO = [p(0, 1), p(0, 2), p(1, 0), p(1, 3), p(2, 1), p(2, 2), PASS, p(1, 1), p(1, 2)]


@pytest.mark.parametrize("rule", ["positional", "situational"])
def test_ko(make_env, rule):
    e = make_env(rule=rule)
    play(e, KO)
    assert e.turn() == 2
    assert e.legal_mask()[p(1, 1)] == 0 and not e.act(p(1, 1)), "immediate recapture recreates the position of two moves ago"
    play(e, [p(4, 4), p(4, 0)])  # a ko threat and its answer: the board differs now
    assert e.legal_mask()[p(1, 1)] == 1
    play(e, [p(1, 1)])           # white retakes; now black may not retake at once
    assert e.legal_mask()[p(1, 2)] == 0 and not e.act(p(1, 2))


def test_positional_superko_is_wider_than_ko(make_env):
    """After black's ko capture white passes and black passes?  No — a pass leaves the position as it is, and only PLAYS are checked: the same ko point stays
    forbidden for white after `white pass, black pass` is impossible (the game is over).  What positional superko adds to the ko rule: white may not retake
    even after a pass pair is avoided by a black pass alone — the position white's capture would create is still the one after move 8."""
    e = make_env(rule="positional")
    play(e, KO + [PASS])         # white passes
    assert not e.is_terminal()
    play(e, [p(4, 4)])           # black plays elsewhere
    # white's recapture now creates a NEW position (a black stone at (4,4) was not there after move 8): legal
    assert e.legal_mask()[p(1, 1)] == 1
    e2 = make_env(rule="positional")
    play(e2, KO + [PASS, PASS])  # white passes, black passes: two passes end the game whatever the ko
    assert e2.is_terminal()


# ---- Tromp-Taylor area ----
def test_area_empty_board_is_the_reference_quirk(make_env):
    """The one place where the reference is NOT the rule text: on an empty board the single empty region has no neighbouring stones at all, and
    `(surrounding & ~black).none()` (go.cpp:713) is vacuously true — the reference gives all N * N points to BLACK (Tromp-Taylor: to nobody).
    A drop-in must reproduce the reference, so both engines are held to 25 here; the test documents the deviation."""
    area_difference_is(make_env, [PASS, PASS], 25)


def test_area_one_stone_owns_the_board(make_env):
    area_difference_is(make_env, [p(2, 2), PASS, PASS], 25)


def test_area_neutral_region(make_env):
    # one stone each: the empty region reaches both colours and counts for nobody
    area_difference_is(make_env, [p(2, 2), p(0, 0), PASS, PASS], 0)


def test_area_wall(make_env):
    # a black wall on column 2 (5 stones), one white stone at (0,4): the left 10 points reach only black, the right 9 empty points reach both
    moves = [p(0, 2), p(0, 4), p(1, 2), PASS, p(2, 2), PASS, p(3, 2), PASS, p(4, 2), PASS, PASS]
    area_difference_is(make_env, moves, 15 - 1)


def test_area_two_territories(make_env):
    # black wall on column 1, white wall on column 3: black 5 + 5, white 5 + 5, the middle column is neutral
    moves = []
    for r in range(5):
        moves += [p(r, 1), p(r, 3)]
    area_difference_is(make_env, moves + [PASS, PASS], 0)
    # ... and with one more black stone inside the neutral column: black + 1, the rest of that column still reaches both
    area_difference_is(make_env, moves + [p(2, 2), PASS, PASS], 1)


# ---- endings ----
def test_two_passes_end_the_game_only_when_consecutive(make_env):
    e = make_env()
    play(e, [PASS, p(0, 0), PASS])
    assert not e.is_terminal()
    play(e, [PASS])
    assert e.is_terminal()


def test_move_cap(make_env):
    """more than 2 * N * N moves end the game (go.cpp:253-254): on 3x3 the game is over after the 19th move and not before, whatever the board.  The moves come
    from a seeded random walk that never passes twice in a row (so two passes cannot end it first); walks that run out of legal plays are discarded."""
    n, cap = 3, 2 * 3 * 3
    reached = 0
    for seed in range(200):
        rng = np.random.default_rng(seed)
        e = make_env(n=n)
        last, count, stuck = -1, 0, False
        while count <= cap:
            assert not e.is_terminal(), f"over after {count} moves without two passes in a row"
            m = e.legal_mask()
            plays = [a for a in range(n * n) if m[a]]
            if last == n * n and not plays:
                stuck = True
                break
            a = int(rng.choice(plays)) if plays and (last == n * n or rng.random() < 0.6) else n * n
            assert e.act(a)
            last = a
            count += 1
        if stuck:
            continue
        assert count == cap + 1 and e.is_terminal()
        reached += 1
        if reached >= 5:
            break
    assert reached >= 1
