"""CPU tests: known-answer tests of the rules engines from OUTSIDE the repo and the reference — the published perft tables of the games (the number of legal move
sequences of every length from the initial position).  The reference holds no fixtures for its environments and its environment sources do not
compile here without Boost, so the oracle's restatement of `environment/othello/othello.cpp:103-262` and `environment/tictactoe/tictactoe.cpp` — and the
product's own host engines beside it — are at least anchored to the games themselves: move generation, flips, the end of the game.
  Othello 8x8 (OEIS A124004; no pass occurs before ply 9): 4, 12, 56, 244, 1396, 8200, 55092
  TicTacToe (sequences that are still running or have just ended at that ply): 9, 72, 504, 3024, 15120, 54720, 148176; games that END at ply 5 / 6 / 7: 1440 / 5328 / 47952
"""
import numpy as np
import pytest


def _perft(env, depth):
    """counts[d] = sequences of d legal moves from the initial position; ended[d] = those after which the game is over.  The engines have no undo: a node is reached by
    reset + replay (what the reference's own record replay does)."""
    counts, ended = [0] * (depth + 1), [0] * (depth + 1)

    def rec(path):
        env.reset()
        for a in path:
            assert env.act(a)
        d = len(path)
        counts[d] += 1
        if env.is_terminal():
            ended[d] += 1
            return
        if d == depth:
            return
        for a in np.nonzero(env.legal_mask())[0]:
            rec(path + [int(a)])

    rec([])
    return counts, ended


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_othello_perft(mz, oracle, which):
    conf = "env_game=othello:env_board_size=8"
    env = oracle.OracleEnv(conf) if which == "oracle" else mz.Env(conf)
    counts, ended = _perft(env, 6)
    assert counts[1:] == [4, 12, 56, 244, 1396, 8200]
    assert sum(ended) == 0  # no game of six plies is over, and no position before ply 9 has a forced pass: every mask above had board moves only
    env.reset()
    assert int(env.legal_mask()[64]) == 0, "the pass (action 64) is only legal where no disc can be placed (ref othello.cpp:180-189)"


@pytest.mark.slow
@pytest.mark.parametrize("which", ["oracle", "product"])
def test_othello_perft_ply_7(mz, oracle, which):
    conf = "env_game=othello:env_board_size=8"
    env = oracle.OracleEnv(conf) if which == "oracle" else mz.Env(conf)
    counts, _ = _perft(env, 7)
    assert counts[7] == 55092


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_tictactoe_perft(mz, oracle, which):
    conf = "env_game=tictactoe"
    env = oracle.OracleEnv(conf) if which == "oracle" else mz.Env(conf)
    counts, ended = _perft(env, 6)
    assert counts[1:] == [9, 72, 504, 3024, 15120, 54720]
    assert ended[:5] == [0, 0, 0, 0, 0] and ended[5] == 1440 and ended[6] == 5328  # three in a row at the earliest with the fifth stone
