"""Hand-computed search cases: what the formulas of the reference's MCTS give on tiny trees, worked out on paper and spelled here in numpy f32 / f64
operations (ref actor/mcts.cpp:20-28 running mean, :40-53 normalized mean, :55-61 PUCT score, :181-203 the arg-max and its tie rule, :205-217 init-Q in both
builds, :163-179 backup, :219-228 value-bound multiset).  Nothing here calls the oracle or the product to GET an expectation — the same cases are then run
against the oracle's tree (tests/test_oracle_search.py, CPU) and against the HIP node pool (tests/test_gpu_pool.py, GPU)."""
import math

import numpy as np

f32 = np.float32


def bias(N, base=19652.0, init=1.25):
    """float puct_bias = init + log((1 + N + base) / base): the ratio in f32, log in double (mcts.cpp:57)"""
    ratio = f32((f32(1) + f32(N) + f32(base)) / f32(base))
    return f32(f32(init) + math.log(float(ratio)))


def u_term(b, policy, N, count):
    """(puct_bias * policy * sqrt(N)) / (1 + count): float * float, then double sqrt promotes the product, then the division (mcts.cpp:58)"""
    return f32((float(f32(b * f32(policy))) * math.sqrt(float(N))) / float(1 + count))


def add(count, mean, v, w=1.0):
    """count += w; mean += w * (v - mean) / count (mcts.cpp:20-28)"""
    count = f32(count + f32(w))
    mean = f32(mean + f32(f32(f32(w) * f32(f32(v) - mean)) / count))
    return count, mean


def case_tie_rule(tree):
    """equal scores: the child with the higher prior wins; equal priors too: the FIRST one (`policy <= best_policy -> continue`, mcts.cpp:191)"""
    t = tree(dict(actor_num_simulation=8))
    t.reset(2)
    assert t.select() == [0]
    t.expand_backup([3, 4, 5], 1, [0.4, 0.4, 0.2], [0.0, 0.0, -0.7], 0.0)
    assert t.select() == [0, 1]  # N = 0: u = 0, init_q = (0 - 1) / (0 + 1) = -1 for all three


def case_flipping_player(tree):
    """a visited child whose action was played by the flipping player ('W' = 2) counts with its mean negated (mcts.cpp:50)"""
    t = tree(dict(actor_num_simulation=8))
    t.reset(1)                                        # the root's action belongs to player 1: its children are white's moves
    t.select()
    t.expand_backup([0, 1], 2, [0.55, 0.45], [0.0, 0.0], 0.0)
    assert t.select() == [0, 1]
    t.expand_backup([2], 1, [1.0], [0.0], 0.8)       # the visited child: mean 0.8, q = -0.8 for white
    # N = 1.  child 0: u = bias(1) * 0.55 * 1 / 2, q = -0.8.  child 1 unvisited: init_q = (-0.8 - 1) / 2 = -0.9, u = bias(1) * 0.45 / 1
    b = bias(1)
    s0 = f32(u_term(b, 0.55, 1, 1) + f32(-0.8))
    s1 = f32(u_term(b, 0.45, 1, 0) + f32(f32(f32(-0.8) - f32(1)) / f32(2)))
    assert s1 > s0                                    # -0.337... against -0.456...
    assert t.select() == [0, 2]


def case_rescale_discount_atari(tree):
    """value rescaling + discount + rewards + the ATARI init-Q on a four-step search; bounds of the multiset after every backup"""
    t = tree(dict(actor_num_simulation=8, actor_mcts_value_rescale=True, actor_mcts_reward_discount=0.5, atari_init_q=True))
    g = f32(0.5)
    t.reset(1)
    assert t.select() == [0]
    t.expand_backup([7, 8, 9], 1, [0.5, 0.3, 0.2], [0.0, 0.0, 0.0], 0.2, 0.0)
    c0, m0 = add(f32(0), f32(0), 0.2)                 # root: count 1, mean 0.2; multiset {0 + g * 0.2}
    assert t.bound() == (1, f32(g * m0), f32(g * m0))
    assert t.root() == (c0, m0)
    # step 1: N = 0 -> u = 0; ATARI init-Q with no visited child = 1 for every child: tie -> highest prior
    assert t.select() == [0, 1]
    t.expand_backup([1, 2], 1, [0.6, 0.4], [0.0, 0.0], 1.0, 2.0)
    c1, m1 = add(f32(0), f32(0), 1.0)                 # node 1: reward 2, mean 1 -> multiset gains 2 + g * 1 = 2.5
    up = f32(f32(2.0) + f32(g * f32(1.0)))            # value handed to the root: r + g * v = 2.5
    old_root = f32(g * m0)
    c0, m0 = add(c0, m0, up)                          # root: count 2, mean 0.2 + (2.5 - 0.2) / 2 = 1.35
    assert m0 == f32(f32(0.2) + f32(f32(up - f32(0.2)) / f32(2)))
    new_root = f32(g * m0)                            # 0.675 replaces 0.1
    assert old_root != new_root
    assert t.bound() == (2, new_root, f32(2.5))
    assert t.root() == (c0, m0)
    # step 2: N = 1.  child 0 (visited): value 2.5 -> (2.5 - lo) / (hi - lo) = 1 -> 2 * 1 - 1 = 1.  ATARI init-Q = mean of the visited = 1
    b = bias(1)
    s = [f32(u_term(b, 0.5, 1, 1) + f32(1)), f32(u_term(b, 0.3, 1, 0) + f32(1)), f32(u_term(b, 0.2, 1, 0) + f32(1))]
    assert int(np.argmax(s)) == 1                     # 1.3125 | 1.375 | 1.25
    assert t.select() == [0, 2]
    t.expand_backup([4], 1, [1.0], [0.0], -1.0, 0.0)
    up2 = f32(f32(0) + f32(g * f32(-1.0)))            # -0.5 (node 2: reward 0, mean -1 -> multiset gains -0.5)
    old_root = f32(g * m0)
    c0, m0 = add(c0, m0, up2)                         # root: count 3
    assert t.bound() == (3, f32(-0.5), f32(2.5))
    assert t.root() == (c0, m0)
    assert f32(g * m0) not in (f32(-0.5), f32(2.5)) and old_root != f32(g * m0)
    # step 3: N = 2.  child 0: q = 1 (its value is the upper bound); child 1: value -0.5 is the lower bound -> 2 * 0 - 1 = -1; init-Q = (1 - 1) / 2 = 0
    b = bias(2)
    s = [f32(u_term(b, 0.5, 2, 1) + f32(1)), f32(u_term(b, 0.3, 2, 1) + f32(-1)), f32(u_term(b, 0.2, 2, 0) + f32(0))]
    assert int(np.argmax(s)) == 0
    # ... and below node 1 (count 1 -> N = 0): u = 0, nothing visited -> init-Q 1 for both -> the higher prior, node 4
    assert t.select() == [0, 1, 4]


def case_rescale_needs_two_bounds(tree):
    """with fewer than two distinct values in the multiset a visited child's normalized mean is 1 (mcts.cpp:44)"""
    t = tree(dict(actor_num_simulation=8, actor_mcts_value_rescale=True))
    t.reset(1)
    t.select()
    t.expand_backup([0, 1], 1, [0.9, 0.1], [0.0, 0.0], 0.0)
    assert t.bound() == (1, f32(0), f32(0))
    assert t.select() == [0, 1]
    t.expand_backup([5], 1, [1.0], [0.0], 0.0)        # value 0 again: the multiset still holds ONE key (0, counted twice ... the root's old 0 left, its new 0 came)
    assert t.bound()[0] == 1
    # N = 1: child 0 visited, q = 1 (size < 2); child 1: init_q (board build) = (1 - 1) / 2 = 0
    b = bias(1)
    s0 = f32(u_term(b, 0.9, 1, 1) + f32(1))
    s1 = f32(u_term(b, 0.1, 1, 0) + f32(0))
    assert s0 > s1
    assert t.select()[:2] == [0, 1]


ALL = [case_tie_rule, case_flipping_player, case_rescale_discount_atari, case_rescale_needs_two_bounds]


class OracleAdapter:
    def __init__(self, oracle, conf):
        self.conf = ":".join(f"{k}={str(v).lower() if isinstance(v, bool) else v}" for k, v in conf.items())
        self.t = oracle.OracleTree(self.conf, 200)

    def reset(self, root_player): self.t.reset(root_player)
    def select(self, start=-1): return [int(x) for x in self.t.select(start)]
    def expand_backup(self, actions, player, policy, logit, value, reward=0.0): self.t.expand_backup(actions, player, policy, logit, value, reward)

    def bound(self):
        import ctypes as C
        lo, hi = C.c_float(), C.c_float()
        n = self.t.L.mzo_tree_value_bound(self.t.h, C.byref(lo), C.byref(hi))
        return (n, f32(lo.value), f32(hi.value))

    def root(self):
        d = self.t.dump()
        return (f32(d["count"][0]), f32(d["mean"][0]))


class PoolAdapter:
    """one game of the HIP node pool through the C ABI"""
    A = 8

    def __init__(self, mz, conf):
        self.pool = mz.Pool(1, 201, self.A, conf.get("actor_num_simulation", 8), reward_discount=conf.get("actor_mcts_reward_discount", 1.0),
                            value_rescale=conf.get("actor_mcts_value_rescale", False), atari_init_q=conf.get("atari_init_q", False))

    def reset(self, root_player): self.pool.reset_search(np.array([root_player], np.int32))

    def select(self, start=-1):
        pl, paths, _ = self.pool.select(None if start < 0 else np.array([start], np.int32))
        return [int(x) for x in paths[0, :pl[0]]]

    def expand_backup(self, actions, player, policy, logit, value, reward=0.0):
        k = len(actions)
        ca, cp, cl = np.zeros((1, self.A), np.int32), np.zeros((1, self.A), np.float32), np.zeros((1, self.A), np.float32)
        ca[0, :k], cp[0, :k], cl[0, :k] = actions, policy, logit
        self.pool.expand_backup(np.array([k], np.int32), ca, cp, cl, np.array([player], np.int32), np.array([value], np.float32), np.array([reward], np.float32))

    def bound(self):
        rr = self.pool.root_read()
        return (int(rr["bound_size"][0]), f32(rr["bound_lo"][0]), f32(rr["bound_hi"][0]))

    def root(self):
        rr = self.pool.root_read()
        return (f32(rr["root_count"][0]), f32(rr["root_mean"][0]))
