"""GPU parity of the search-pool kernels (select / expand / backup, through the C ABI) against the CPU
oracle's MCTS on identical injected network outputs.  Integer and f32 tree state must be BIT-EXACT."""
import numpy as np
import pytest

from helpers import bits

pytestmark = pytest.mark.gpu

NODE_KEYS = ["action", "player", "num_children", "first_child", "mean", "count", "policy", "logit", "noise", "value", "reward"]


def make_candidates(rng, A, allow_terminal, tie_prob):
    """A candidate list as the actor would hand it over: sorted by prior descending, with occasional exact ties."""
    if allow_terminal and rng.random() < 0.05:
        return np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros(0, np.float32)
    k = int(rng.integers(1, A + 1))
    actions = rng.permutation(A)[:k].astype(np.int32)
    logits = rng.normal(0, 2, k).astype(np.float32)
    if rng.random() < tie_prob and k > 2:
        logits[rng.integers(0, k, size=max(2, k // 3))] = logits[0]  # exact ties in prior
    p = np.exp(logits - logits.max()).astype(np.float32)
    p = (p / p.sum()).astype(np.float32)
    order = np.argsort(-p, kind="stable")
    return actions[order], p[order], logits[order]


def drive(mz, oracle, G, A, n, conf, pool_kw, seed, gumbel=False, use_reward=False, oracle_games=None, noise=True):
    rng = np.random.default_rng(seed)
    cap = 1 + (n + 1) * A
    pool = mz.Pool(G, cap, A, n, **pool_kw)
    og = list(range(G)) if oracle_games is None else list(oracle_games)
    trees = {g: oracle.OracleTree(conf, (n + 1) * A) for g in og}
    root_player = rng.integers(1, 3, G).astype(np.int32)
    pool.reset_search(root_player)
    for g in og:
        trees[g].reset(int(root_player[g]))
    node_player = [{0: int(root_player[g])} for g in range(G)]
    num_nodes = np.ones(G, np.int64)
    for sim in range(n + 1):
        start = None
        if gumbel and sim >= 1:
            # a root child, as GumbelZero::selection would pass it (ref gumbel_zero.cpp:80-85)
            rr = pool.root_read()
            start = np.array([1 + int(rng.integers(0, rr["num_children"][g])) for g in range(G)], np.int32)
        plen, paths, pacts = pool.select(start)
        for g in og:
            op = trees[g].select(-1 if start is None else int(start[g]))
            assert plen[g] == len(op) and np.array_equal(paths[g, :plen[g]], op), f"sim {sim} game {g}: path {paths[g, :plen[g]]} != {op}"
        cc = np.zeros(G, np.int32)
        ca = np.zeros((G, A), np.int32)
        cp = np.zeros((G, A), np.float32)
        cl = np.zeros((G, A), np.float32)
        pl = np.zeros(G, np.int32)
        val = rng.uniform(-1, 1, G).astype(np.float32)
        rew = (rng.uniform(0, 1, G) < 0.3).astype(np.float32) * rng.uniform(0, 2, G).astype(np.float32) if use_reward else np.zeros(G, np.float32)
        for g in range(G):
            leaf = int(paths[g, plen[g] - 1])
            a, p, l = make_candidates(rng, A, allow_terminal=(sim > 0), tie_prob=0.3)
            k = len(a)
            cc[g] = k
            ca[g, :k], cp[g, :k], cl[g, :k] = a, p, l
            pl[g] = 3 - node_player[g][leaf] if not use_reward else node_player[g][leaf]  # two-player vs single-player
            for i in range(k):
                node_player[g][int(num_nodes[g]) + i] = int(pl[g])
            num_nodes[g] += k
            if g in trees:
                trees[g].expand_backup(a, int(pl[g]), p, l, float(val[g]), float(rew[g]))
        pool.expand_backup(cc, ca, cp, cl, pl, val, rew)
        if sim == 0 and noise:  # root noise (ref zero_actor.cpp:194-213): host computes, device stores
            npol = np.zeros((G, A), np.float32)
            nlog = np.zeros((G, A), np.float32)
            nnoi = np.zeros((G, A), np.float32)
            for g in range(G):
                k = cc[g]
                nz = rng.gamma(0.3, 1.0, k).astype(np.float32)
                nnoi[g, :k] = nz
                npol[g, :k] = (np.float32(0.75) * cp[g, :k] + np.float32(0.25) * nz).astype(np.float32)
                nlog[g, :k] = cl[g, :k] + nz
                if g in trees:
                    for i in range(k):
                        trees[g].set_child_policy(1 + i, npol[g, i], nlog[g, i], nnoi[g, i])
            pool.root_set_noise(npol, nlog, nnoi)
    return pool, trees, num_nodes


def compare_trees(pool, trees, num_nodes):
    for g, t in trees.items():
        assert pool.num_nodes(g) == t.num_nodes() == num_nodes[g]
        a, b = pool.read_nodes(g), t.dump()
        for k in NODE_KEYS:
            if a[k].dtype == np.float32:
                assert np.array_equal(bits(a[k]), bits(b[k])), f"game {g}: {k} differs at {np.nonzero(bits(a[k]) != bits(b[k]))[0][:5]}"
            else:
                assert np.array_equal(a[k], b[k]), f"game {g}: {k} differs"


def test_puct_search_bit_exact(mz, oracle):
    pool, trees, nn = drive(mz, oracle, G=24, A=20, n=80, conf="actor_num_simulation=80", pool_kw={}, seed=1)
    compare_trees(pool, trees, nn)
    rr = pool.root_read()
    assert np.all(rr["root_count"] == 81)
    for g, t in trees.items():
        d = t.dump()
        k = d["num_children"][0]
        assert rr["num_children"][g] == k
        assert np.array_equal(bits(rr["count"][g, :k]), bits(d["count"][1:1 + k])) and np.array_equal(rr["action"][g, :k], d["action"][1:1 + k])
        assert np.array_equal(bits(rr["mean"][g, :k]), bits(d["mean"][1:1 + k]))


def test_wide_nodes_two_chunks(mz, oracle):
    """A = 82 (9x9 Go): children span two 64-lane chunks"""
    pool, trees, nn = drive(mz, oracle, G=8, A=82, n=60, conf="actor_num_simulation=60", pool_kw={}, seed=2)
    compare_trees(pool, trees, nn)


def test_value_rescale_discount_atari_initq(mz, oracle):
    conf = ("actor_num_simulation=50:actor_mcts_value_rescale=true:actor_mcts_reward_discount=0.997:atari_init_q=true:"
            "actor_mcts_value_flipping_player=W")
    kw = dict(value_rescale=True, reward_discount=0.997, atari_init_q=True, flipping_player=2)
    pool, trees, nn = drive(mz, oracle, G=16, A=18, n=50, conf=conf, pool_kw=kw, seed=3, use_reward=True)
    compare_trees(pool, trees, nn)
    rr = pool.root_read()
    for g, t in trees.items():
        import ctypes as C
        lo, hi = C.c_float(), C.c_float()
        size = t.L.mzo_tree_value_bound(t.h, C.byref(lo), C.byref(hi))
        assert rr["bound_size"][g] == size
        assert np.float32(lo.value) == rr["bound_lo"][g] and np.float32(hi.value) == rr["bound_hi"][g]


def test_gumbel_start_nodes_and_puct_params(mz, oracle):
    conf = "actor_num_simulation=32:actor_mcts_puct_base=100:actor_mcts_puct_init=2.5:actor_mcts_value_flipping_player=B"
    kw = dict(puct_base=100.0, puct_init=2.5, flipping_player=1)
    pool, trees, nn = drive(mz, oracle, G=16, A=65, n=32, conf=conf, pool_kw=kw, seed=4, gumbel=True)
    compare_trees(pool, trees, nn)


def test_baseline_size_c2_properties(mz, oracle):
    """BASELINE.json configs[1] sizes: 256 games x (1 + 401*82) nodes, n = 400.  Size-independent invariants on
    every game + bit-exact oracle comparison on 3 games."""
    G, A, n = 256, 82, 400
    pool, trees, nn = drive(mz, oracle, G=G, A=A, n=n, conf="actor_num_simulation=400", pool_kw={}, seed=5, oracle_games=[0, 100, 255])
    compare_trees(pool, trees, nn)
    rr = pool.root_read()
    assert np.all(rr["root_count"] == n + 1)
    for g in range(G):
        k = rr["num_children"][g]
        assert rr["count"][g, :k].sum() == n  # every simulation after the root expansion went through exactly one root child
        assert pool.num_nodes(g) == nn[g]


def test_capacity_and_argument_errors(mz):
    pool = mz.Pool(2, 5, 4, 4)
    pool.reset_search(np.array([2, 2], np.int32))
    pool.select()
    cc = np.array([4, 4], np.int32)
    z = np.zeros((2, 4), np.float32)
    pool.expand_backup(cc, np.zeros((2, 4), np.int32), z, z, np.array([1, 1], np.int32), np.zeros(2, np.float32))
    pool.select()
    with pytest.raises(mz.MzError):  # 1 + 4 + 4 > 5 nodes
        pool.expand_backup(cc, np.zeros((2, 4), np.int32), z, z, np.array([2, 2], np.int32), np.zeros(2, np.float32))
    with pytest.raises(mz.MzError):
        pool.expand_backup(np.array([9, 0], np.int32), np.zeros((2, 4), np.int32), z, z, np.array([2, 2], np.int32), np.zeros(2, np.float32))
    with pytest.raises(mz.MzError):
        mz.Pool(0, 5, 4, 4)


@pytest.mark.parametrize("case", __import__("hand_cases").ALL, ids=lambda c: c.__name__)
def test_hand_computed_search_cases(mz, case):
    """tests/hand_cases.py: the expectations come from the reference's formulas worked out by hand (numpy f32 / f64), not from the oracle"""
    import hand_cases
    case(lambda conf: hand_cases.PoolAdapter(mz, conf))
