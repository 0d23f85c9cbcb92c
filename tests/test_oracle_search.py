"""CPU known-answer tests of the oracle's search/actor restatement (PARITY UNPINNED part: these are facts the
survey verified on the reference run with its own probe, hand-computed formula cases, and invariants)."""
import math
import re

import numpy as np
import pytest

LINE = re.compile(r"^SelfPlay (true|false) (\d+) (\d+) (-?[0-9.]+) (\(;GM\[[a-z0-9_x]+\]RE\[[^\]]+\]OBS\[\]SZ\[\d+\](KM\[[0-9.]+\])?EV\[[^\]]*\]DLEN\[\d+-\d+\](;[BW]\[\d+\]P\[[^\]]*\]V\[-?\d+\.\d{6}\]R\[[^\]]*\])*\)) #$")


def test_puct_formula_hand_case(oracle):
    """one expanded root with 3 children, one visited child: check the PUCT pick by hand (mcts.cpp:55-61,181-217)"""
    t = oracle.OracleTree("actor_num_simulation=10", 100)
    t.reset(2)
    assert list(t.select()) == [0]
    t.expand_backup([5, 6, 7], 1, [0.5, 0.3, 0.2], [1.0, 0.5, 0.1], 0.2)
    # N = 0 -> sqrt(0) = 0 -> every u = 0; no visited child -> init_q = (0 - 1) / (0 + 1) = -1 for all: tie -> highest prior = child 0
    assert list(t.select()) == [0, 1]
    t.expand_backup([1, 2], 2, [0.6, 0.4], [0, 0], -0.4)  # child value -0.4 from player-2's view; stored mean = -0.4
    d = t.dump()
    assert d["count"][0] == 2 and d["count"][1] == 1 and np.float32(d["mean"][1]) == np.float32(-0.4)
    # now N = 1: bias = 1.25 + log((1+1+19652)/19652); visited child q = mean (player 1 = 'B', flipping player is 'W') = -0.4
    bias = np.float32(1.25 + math.log(float(np.float32((1 + 1 + np.float32(19652)) / np.float32(19652)))))
    init_q = np.float32((np.float32(-0.4) - 1) / 2)
    s0 = np.float32(np.float32(bias * np.float32(0.5)) * 1.0 / 2.0) + np.float32(-0.4)
    s1 = np.float32(np.float32(bias * np.float32(0.3)) * 1.0 / 1.0) + init_q
    s2 = np.float32(np.float32(bias * np.float32(0.2)) * 1.0 / 1.0) + init_q
    best = int(np.argmax([s0, s1, s2]))
    assert list(t.select())[1] == 1 + best


def test_running_mean_and_backup_signs(oracle):
    t = oracle.OracleTree("actor_num_simulation=10:actor_mcts_reward_discount=0.5", 100)
    t.reset(2)
    t.select()
    t.expand_backup([0, 1], 1, [0.7, 0.3], [0, 0], 0.5, 0.0)
    t.select()
    t.expand_backup([0], 2, [1.0], [0], 1.0, 2.0)  # leaf reward 2, value 1
    d = t.dump()
    # leaf: count 1 mean 1; root: count 2, mean = 0.5 + (r + g*v - 0.5)/2 with r + g*v = 2 + 0.5*1 = 2.5
    assert d["reward"][1] == 2.0 and d["value"][1] == 1.0 and d["mean"][1] == 1.0
    assert d["count"][0] == 2 and d["mean"][0] == np.float32(0.5 + (2.5 - 0.5) / 2)


def _weights(oracle, desc):
    return oracle.gen_weights(desc, 0)


def test_selfplay_lines_format_and_determinism(oracle):
    d = oracle.desc_c1()
    w = _weights(oracle, d)
    conf = "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=8:zero_num_threads=1:program_seed=1:nn_file_name=/x/ttt.pt"
    runs = []
    for _ in range(2):
        g = oracle.OracleGroup(conf, d, w)
        g.cycles(17 * 30)
        runs.append(g.lines())
        assert g.leaf_evals() == 17 * 30 * 8
    assert runs[0] == runs[1] and len(runs[0]) > 20
    for l in runs[0]:
        m = LINE.match(l)
        assert m, l
        moves = l.count(";B[") + l.count(";W[")
        assert int(m.group(2)) == moves == int(m.group(3)) and "EV[ttt.pt]" in l and f"DLEN[0-{moves - 1}]" in l
        # every move's visit counts sum to n (the root evaluation itself is not a child visit)
        for p in re.findall(r"P\[([^\]]*)\]", l):
            assert sum(int(x.split(":")[1]) for x in p.split(",")) == 16
    other = oracle.OracleGroup(conf.replace("program_seed=1", "program_seed=2"), d, w)
    other.cycles(17 * 30)
    assert other.lines() != runs[0]


def test_gumbel_visit_pattern(oracle):
    """SURVEY.md A11 (verified by the survey on the compiled reference): n=16, m=16 -> 16 candidates visited once each;
    n=50, m=16 -> root child counts are 8x1 + 4x2 + 2x5 + 2x12."""
    d = oracle.make_desc("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1)
    w = _weights(oracle, d)
    base = ("env_game=go:env_board_size=9:actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:"
            "actor_gumbel_sample_size=16:zero_num_parallel_games=3:program_seed=5:actor_num_simulation=")
    for n, expect in ((16, [1] * 16), (50, [1] * 8 + [2] * 4 + [5] * 2 + [12] * 2)):
        g = oracle.OracleGroup(base + str(n), d, w)
        g.set_trace(True)
        g.cycles((n + 1) * 3 + 1)  # the results of cycle k are applied in cycle k+1
        rows = [t for t in g.trace() if t.startswith("R ")]
        assert len(rows) == 9
        for t in rows:
            counts = sorted(int(float(c)) for c in t.split("counts=")[1].split(",") if float(c) > 0)
            assert counts == expect, t


def test_go_game_reaches_move_cap_and_scores(oracle):
    d = oracle.make_desc("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1)
    w = _weights(oracle, d)
    g = oracle.OracleGroup("env_game=go:env_board_size=9:actor_num_simulation=4:zero_num_parallel_games=2:program_seed=3:nn_file_name=g.pt", d, w)
    g.cycles(5 * 170)
    lines = g.lines()
    assert lines
    for l in lines:
        assert LINE.match(l), l
        assert "KM[7.500000]" in l and "GM[go_9x9]" in l and "SZ[9]" in l


def test_group_commands_between_cycles(oracle):
    """Group::command (ref actor_group.cpp:200-252) on the CPU: `stop` gates the cycles, an ignored reset_actors and a load_model of the SAME weights
    change nothing, a load_model of other weights changes the records from that cycle on (and the EV tag), reset_actors outside the ignore list drops
    the games in flight, update_config is seen by the next decision."""
    conf = "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=4:program_seed=3:nn_file_name=/m/weight_iter_0.pt:zero_num_threads=1"
    od = oracle.desc_c1()
    w0, w1 = oracle.gen_weights(od, 0), oracle.gen_weights(od, 1)

    def play(schedule, extra=""):
        og = oracle.OracleGroup(conf + extra, od, w0)
        for step in schedule:
            if isinstance(step, int):
                og.cycles(step)
            else:
                og.command(*step)
        return og.lines(), og.peek_records(4), og.num_cycles()

    base = play([17 * 30])
    assert play([17 * 10 + 5, ("stop",), 50, ("start",), 17 * 20 - 5]) == base
    assert play([17 * 10 + 5, ("reset_actors",), ("load_model /m/weight_iter_0.pt", w0), ("keep_alive",), 17 * 20 - 5]) == base
    swapped = play([17 * 10 + 5, ("load_model /m/weight_iter_1.pt", w1), 17 * 20 - 5])
    k = next(i for i, (a, b) in enumerate(zip(base[0], swapped[0])) if a != b)
    assert k > 0 and all("EV[weight_iter_0.pt]" in l for l in swapped[0][:k]) and "EV[weight_iter_1.pt]" in swapped[0][-1]
    reset = play([17 * 10 + 5, ("reset_actors",), 17 * 20 - 5], ":zero_actor_ignored_command=keep_alive")
    assert reset[0][:len(base[0]) // 4] == base[0][:len(base[0]) // 4] and reset[0] != base[0] and reset[2] == base[2]
    hot = play([17 * 10, ("update_config actor_select_action_softmax_temperature=0.05",), 17 * 20])
    assert hot[0] != base[0] and hot[0][:3] == base[0][:3]


@pytest.mark.parametrize("case", __import__("hand_cases").ALL, ids=lambda c: c.__name__)
def test_hand_computed_search_cases(oracle, case):
    """tests/hand_cases.py: paper-and-pencil PUCT / init-Q / backup / value-bound cases, here against the oracle's tree (the HIP pool: tests/test_gpu_pool.py)"""
    import hand_cases
    case(lambda conf: hand_cases.OracleAdapter(oracle, conf))
