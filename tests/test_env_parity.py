"""CPU tests: the product's host rules engines (libmzgpu, own design) against the oracle's restatement of the
reference envs, in the style of the reference's `-mode env_test` (ref console/mode_handler.cpp:167-192):
random legal playouts to the end, comparing at every ply the legal mask, turn, terminal flag, scores and the
feature planes under all 8 rotations; plus replay-from-record determinism."""
import numpy as np
import pytest

GAMES = [("env_game=tictactoe", 60), ("env_game=othello:env_board_size=8", 25), ("env_game=go:env_board_size=9", 12),
         ("env_game=go:env_board_size=5", 25), ("env_game=othello:env_board_size=6", 20), ("env_game=go:env_board_size=13", 3),
         ("env_game=go:env_board_size=5:env_go_ko_rule=situational", 25), ("env_game=go:env_board_size=4:env_go_ko_rule=situational", 40),
         ("env_game=go:env_board_size=9:env_go_ko_rule=situational", 8)]


@pytest.mark.parametrize("conf,playouts", GAMES)
def test_random_playouts_match_oracle(mz, oracle, conf, playouts):
    rng = np.random.default_rng(hash(conf) % (1 << 32))
    a, b = mz.Env(conf), oracle.OracleEnv(conf)
    total_moves = 0
    for game in range(playouts):
        a.reset()
        b.reset()
        history = []
        while True:
            ma, mb = a.legal_mask(), b.legal_mask()
            assert np.array_equal(ma, mb), f"{conf}: legal mask differs after {history}"
            assert a.turn() == b.turn() and a.is_terminal() == b.is_terminal()
            assert a.eval_score() == b.eval_score() and a.eval_score(True) == b.eval_score(True)
            if len(history) % 7 == 0 or a.is_terminal():
                for rot in range(8):
                    assert np.array_equal(a.features(rot), b.features(rot)), f"{conf}: features rot {rot} differ after {history}"
            else:
                rot = int(rng.integers(0, 8))
                assert np.array_equal(a.features(rot), b.features(rot))
            if a.is_terminal():
                break
            legal = np.nonzero(ma)[0]
            # bias towards board moves so that Go games are not all immediate double passes
            if len(legal) > 1 and rng.random() < 0.97:
                legal = legal[legal != a.policy_size() - 1] if "tictactoe" not in conf else legal
            act = int(rng.choice(legal))
            # an illegal action must be refused by both and change nothing
            illegal = np.nonzero(ma == 0)[0]
            if len(illegal):
                bad = int(rng.choice(illegal))
                assert not a.act(bad) and not b.act(bad)
            assert a.act(act) and b.act(act)
            history.append(act)
        total_moves += len(history)
        # replay determinism (record round trip of env_test): same actions -> same final state
        a.reset()
        for act in history:
            assert a.act(act)
        assert a.is_terminal() and a.eval_score() == b.eval_score() and np.array_equal(a.features(0), b.features(0))
    assert total_moves > playouts * 4


def test_go_superko_and_capture_cases(mz, oracle):
    """hand-made 5x5 positions: ko recapture is illegal (positional superko), suicide illegal, capture legal"""
    conf = "env_game=go:env_board_size=5"
    for env in (mz.Env(conf), oracle.OracleEnv(conf)):
        # B: 1,5,11,7  W: 2,8,12  then W takes at 6?  build a ko around points 6/7
        seq = [(1, 1), (2, 2), (5, 1), (8, 2), (11, 1), (12, 2), (7, 1), (6, 2)]  # W 6 captures B 7
        for a, p in seq:
            assert env.act(a, p), (a, p)
        m = env.legal_mask()
        assert m[7] == 0, "immediate ko recapture must be illegal (positional superko)"
        assert env.act(24, 1) and env.act(23, 2)
        assert env.legal_mask()[7] == 1, "after a ko threat elsewhere the position is new: recapture legal"
    e = mz.Env(conf)
    for a, p in [(1, 1), (24, 2), (5, 1), (23, 2)]:
        assert e.act(a, p)
    assert e.legal_mask()[0] == 1 and e.turn() == 1  # own eye: legal for black
    e.act(22, 1)
    assert e.legal_mask()[0] == 0  # suicide for white


def test_situational_superko_differs_from_positional(mz, oracle):
    """env_go_ko_rule=situational (ref go.cpp:45-49,141,222): a position only repeats with the same player to move.  Same random 3x3
    playouts under both rules on the host engine and on the oracle: the engines agree under each rule, and the rules do differ."""
    rng = np.random.default_rng(5)
    differ = 0
    for game in range(120):
        envs = {r: (mz.Env(f"env_game=go:env_board_size=3:env_go_ko_rule={r}"), oracle.OracleEnv(f"env_game=go:env_board_size=3:env_go_ko_rule={r}"))
                for r in ("positional", "situational")}
        for ply in range(60):
            masks = {}
            for r, (a, b) in envs.items():
                masks[r] = a.legal_mask()
                assert np.array_equal(masks[r], b.legal_mask()), (r, game, ply)
            differ += int(not np.array_equal(masks["positional"], masks["situational"]))
            both = np.nonzero(masks["positional"] & masks["situational"])[0]
            board = both[both != 9]
            act = 9 if len(board) == 0 or rng.random() < 0.2 else int(rng.choice(board))
            for a, b in envs.values():
                assert a.act(act) and b.act(act)
            if envs["positional"][0].is_terminal():
                break
    assert differ > 0, "the playouts never reached a position where the two rules differ"
    with pytest.raises(mz.MzError):
        mz.Env("env_game=go:env_go_ko_rule=natural")


def test_tromp_taylor_empty_board_goes_to_black(mz, oracle):
    """ref go.cpp:714: an empty region with an empty border counts for black"""
    conf = "env_game=go:env_board_size=9"
    for env in (mz.Env(conf), oracle.OracleEnv(conf)):
        assert env.act(81) and env.act(81) and env.is_terminal()
        assert env.eval_score() == 1.0


def test_env_shapes(mz):
    for conf, A, F in [("env_game=tictactoe", 9, 36), ("env_game=othello", 65, 256), ("env_game=go", 82, 18 * 81)]:
        e = mz.Env(conf)
        assert e.policy_size() == A and e.features(0).size == F
    with pytest.raises(mz.MzError):
        mz.Env("env_game=chess")
    with pytest.raises(mz.MzError):
        mz.Env("env_game=othello:env_board_size=10")


def test_atari_shaped_env_matches_oracle(mz, oracle):
    conf = "env_game=atari:env_atari_episode_length=12"
    a, b = mz.Env(conf), oracle.OracleEnv(conf)
    # both sides take the seed from their caller's RNG stream; pin it to the oracle's for the comparison
    rng = np.random.default_rng(3)
    for episode in range(2):
        b.reset()
        seed = b.seed()
        a.reset_seed(seed)
        steps = 0
        while True:
            assert a.is_terminal() == b.is_terminal() and a.eval_score() == b.eval_score()
            assert np.array_equal(a.legal_mask(), b.legal_mask()) and a.legal_mask().sum() == 18
            fa, fb = a.features(0), b.features(0)
            assert fa.size == 32 * 96 * 96 and np.array_equal(fa, fb), f"features differ at step {steps}"
            if a.is_terminal():
                break
            act = int(rng.integers(0, 18))
            assert a.act(act) and b.act(act)
            assert a.reward() == b.reward()
            steps += 1
        assert steps == 12


def test_action_strings_match_the_reference_conversion(mz):
    """BaseEnv::act(const std::vector<std::string>&) reads its second argument through SGFLoader::boardCoordinateStringToActionID
    (ref base_env.h:326-333, utils/sgf_loader.cpp:89-99): the product's conversion (mz_env_action_from_string, behind
    mz_worker_act_string / BaseActor::act(vector<string>)) against the outputs of the reference's own function compiled in place
    (tests/golden/ref_sgf_vectormap.json, made by gen_ref_sgf_golden.py)."""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sgf_vectormap.json")))
    envs = {3: mz.Env("env_game=tictactoe"), 9: mz.Env("env_game=go:env_board_size=9"), 13: mz.Env("env_game=go:env_board_size=13"),
            19: mz.Env("env_game=go:env_board_size=19")}
    checked = 0
    for e in fx["coords"]:
        assert envs[e["n"]].action_from_string(e["coord"]) == e["out"][0], e
        checked += 1
    assert checked == 40
    # the Atari-shaped game: ALE's action names without the PLAYER_A_ prefix, any case (ref atari.cpp:9-39); unknown -> -1
    at = mz.Env("env_game=atari")
    assert [at.action_from_string(s) for s in ("NOOP", "fire", "Up", "DOWNLEFTFIRE", "jump")] == [0, 1, 2, 17, -1]
