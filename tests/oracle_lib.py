"""ctypes bindings for oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class NetDesc(C.Structure):
    """Same field order as oracle/oracle.h NetDesc and include/mzgpu.h mz_net_desc."""
    _fields_ = [("game_name", C.c_char * 64),
                ("num_input_channels", C.c_int), ("input_channel_height", C.c_int), ("input_channel_width", C.c_int),
                ("num_hidden_channels", C.c_int), ("hidden_channel_height", C.c_int), ("hidden_channel_width", C.c_int),
                ("num_action_feature_channels", C.c_int), ("num_blocks", C.c_int), ("action_size", C.c_int),
                ("num_value_hidden_channels", C.c_int), ("discrete_value_size", C.c_int), ("type", C.c_int)]


NET_TYPES = {"alphazero": 0, "muzero": 1, "muzero_atari": 2}


def make_desc(game, cin, h, w, ch, hh, hw, ac, blocks, actions, vh=256, dv=1, type_name="alphazero"):
    """Argument order of the reference's create_network (network/py/create_network.py:6-18)."""
    d = NetDesc()
    d.game_name = game.encode()
    (d.num_input_channels, d.input_channel_height, d.input_channel_width) = (cin, h, w)
    (d.num_hidden_channels, d.hidden_channel_height, d.hidden_channel_width) = (ch, hh, hw)
    (d.num_action_feature_channels, d.num_blocks, d.action_size) = (ac, blocks, actions)
    (d.num_value_hidden_channels, d.discrete_value_size, d.type) = (vh, dv, NET_TYPES[type_name])
    return d


# the five BASELINE.json configs
def desc_c1(): return make_desc("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9)
def desc_c2(): return make_desc("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82)
def desc_c3(): return make_desc("othello_8x8", 4, 8, 8, 64, 8, 8, 1, 6, 65)
def desc_c4(): return make_desc("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, type_name="muzero")
def desc_c5(): return make_desc("atari_ms_pacman", 32, 96, 96, 64, 6, 6, 18, 6, 18, 256, 601, type_name="muzero_atari")


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    L.mzo_net_param_count.restype = C.c_long
    L.mzo_net_param_count.argtypes = [C.POINTER(NetDesc)]
    L.mzo_net_generate.argtypes = [C.POINTER(NetDesc), C.c_ulonglong, fp]
    L.mzo_net_create.restype = C.c_void_p
    L.mzo_net_create.argtypes = [C.POINTER(NetDesc), fp, C.c_long]
    L.mzo_net_destroy.argtypes = [C.c_void_p]
    L.mzo_net_forward_az.argtypes = [C.c_void_p, fp, C.c_int, fp, fp, fp]
    L.mzo_net_initial.argtypes = [C.c_void_p, fp, C.c_int, fp, fp, fp, fp]
    L.mzo_net_recurrent.argtypes = [C.c_void_p, fp, fp, C.c_int, fp, fp, fp, fp, fp]
    L.mzo_invert_value.restype = C.c_float
    L.mzo_invert_value.argtypes = [C.c_float]
    L.mzo_transform_value.restype = C.c_float
    L.mzo_transform_value.argtypes = [C.c_float]
    L.mzo_expf.argtypes = [fp, C.c_int, fp]
    L.mzo_tanhf.argtypes = [fp, C.c_int, fp]
    L.mzo_rng_vector.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_double)]
    L.mzo_config_dump.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.mzo_env_create.restype = C.c_void_p
    L.mzo_env_create.argtypes = [C.c_char_p]
    for name in ("mzo_env_destroy", "mzo_env_reset"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.mzo_env_act.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for name in ("mzo_env_turn", "mzo_env_is_terminal", "mzo_env_policy_size", "mzo_env_num_input_channels", "mzo_env_board_size"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.mzo_env_eval_score.restype = C.c_float
    L.mzo_env_eval_score.argtypes = [C.c_void_p, C.c_int]
    L.mzo_env_reward.restype = C.c_float
    L.mzo_env_reward.argtypes = [C.c_void_p]
    L.mzo_env_seed.argtypes = [C.c_void_p]
    L.mzo_env_legal_mask.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte)]
    L.mzo_env_features.argtypes = [C.c_void_p, C.c_int, fp]
    L.mzo_env_action_features.argtypes = [C.c_void_p, C.c_int, C.c_int, fp]
    L.mzo_tree_create.restype = C.c_void_p
    L.mzo_tree_create.argtypes = [C.c_char_p, C.c_long]
    L.mzo_tree_destroy.argtypes = [C.c_void_p]
    L.mzo_tree_reset.argtypes = [C.c_void_p, C.c_int]
    L.mzo_tree_select.argtypes = [C.c_void_p, C.c_int, ip, C.c_int]
    L.mzo_tree_expand_backup.argtypes = [C.c_void_p, C.c_int, ip, C.c_int, fp, fp, C.c_float, C.c_float]
    L.mzo_tree_set_child_policy.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]
    L.mzo_tree_num_nodes.argtypes = [C.c_void_p]
    L.mzo_tree_dump.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, ip, fp, fp, fp, fp, fp, fp, fp]
    L.mzo_tree_value_bound.argtypes = [C.c_void_p, fp, fp]
    L.mzo_compress_string.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.mzo_sgf_parse.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.mzo_tagmap_apply.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.mzo_sgf_coords.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    L.mzo_sgf_strings.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.mzo_loader_create.restype = C.c_void_p
    L.mzo_loader_create.argtypes = [C.c_char_p]
    L.mzo_loader_destroy.argtypes = [C.c_void_p]
    L.mzo_loader_add.argtypes = [C.c_void_p, C.c_char_p]
    L.mzo_loader_finish.argtypes = [C.c_void_p]
    L.mzo_loader_load_file.argtypes = [C.c_void_p, C.c_char_p]
    L.mzo_loader_num_data.argtypes = [C.c_void_p]
    L.mzo_loader_num_games.argtypes = [C.c_void_p]
    L.mzo_loader_sample.argtypes = [C.c_void_p, fp, fp, fp, fp, fp, fp, ip]
    L.mzo_loader_update_priority.argtypes = [C.c_void_p, ip, fp]
    L.mzo_group_create.restype = C.c_void_p
    L.mzo_group_create.argtypes = [C.c_char_p, C.POINTER(NetDesc), fp, C.c_long]
    L.mzo_group_destroy.argtypes = [C.c_void_p]
    L.mzo_group_set_trace.argtypes = [C.c_void_p, C.c_int]
    L.mzo_group_cycles.argtypes = [C.c_void_p, C.c_int]
    L.mzo_group_command.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_long]
    L.mzo_group_num_cycles.restype = C.c_ulonglong
    L.mzo_group_num_cycles.argtypes = [C.c_void_p]
    L.mzo_group_leaf_evals.restype = C.c_ulonglong
    L.mzo_group_leaf_evals.argtypes = [C.c_void_p]
    L.mzo_group_games.restype = C.c_ulonglong
    L.mzo_group_games.argtypes = [C.c_void_p]
    L.mzo_group_num_lines.argtypes = [C.c_void_p]
    L.mzo_group_line.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    L.mzo_group_peek_record.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    L.mzo_group_num_trace.argtypes = [C.c_void_p]
    L.mzo_group_trace.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    _lib = L
    return L


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int))


def gen_weights(desc, seed=0):
    L = lib()
    n = L.mzo_net_param_count(C.byref(desc))
    w = np.empty(n, np.float32)
    L.mzo_net_generate(C.byref(desc), seed, fptr(w))
    return w


class OracleNet:
    def __init__(self, desc, raw):
        self.L = lib()
        self.desc = desc
        self.raw = np.ascontiguousarray(raw, np.float32)
        self.h = self.L.mzo_net_create(C.byref(desc), fptr(self.raw), self.raw.size)
        assert self.h, "oracle net create failed"

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mzo_net_destroy(self.h)
            self.h = None

    def forward_az(self, feat):
        feat = np.ascontiguousarray(feat, np.float32)
        B, A = feat.shape[0], self.desc.action_size
        p, l, v = np.empty((B, A), np.float32), np.empty((B, A), np.float32), np.empty(B, np.float32)
        self.L.mzo_net_forward_az(self.h, fptr(feat), B, fptr(p), fptr(l), fptr(v))
        return p, l, v

    def _hs(self):
        d = self.desc
        return d.num_hidden_channels * d.hidden_channel_height * d.hidden_channel_width

    def invert(self, a):
        return np.array([self.L.mzo_invert_value(float(v)) for v in np.asarray(a).reshape(-1)], np.float32)

    def initial(self, feat):
        feat = np.ascontiguousarray(feat, np.float32)
        B, A = feat.shape[0], self.desc.action_size
        p, l, v = np.empty((B, A), np.float32), np.empty((B, A), np.float32), np.empty(B, np.float32)
        h = np.empty((B, self._hs()), np.float32)
        self.L.mzo_net_initial(self.h, fptr(feat), B, fptr(p), fptr(l), fptr(v), fptr(h))
        return p, l, v, h

    def recurrent(self, hidden, action):
        hidden = np.ascontiguousarray(hidden, np.float32)
        action = np.ascontiguousarray(action, np.float32)
        B, A = hidden.shape[0], self.desc.action_size
        p, l, v = np.empty((B, A), np.float32), np.empty((B, A), np.float32), np.empty(B, np.float32)
        r, h = np.zeros(B, np.float32), np.empty((B, self._hs()), np.float32)
        self.L.mzo_net_recurrent(self.h, fptr(hidden), fptr(action), B, fptr(p), fptr(l), fptr(v), fptr(r), fptr(h))
        return p, l, v, r, h


def compress_string(data):
    """compressString (ref utils/utils.h:35-91) of the oracle: gzip member, lower-case hex"""
    L = lib()
    data = bytes(data)
    n = L.mzo_compress_string(data, len(data), None, 0)
    buf = C.create_string_buffer(n + 1)
    L.mzo_compress_string(data, len(data), buf, n + 1)
    return buf.value.decode()


class OracleLoader:
    """The oracle's learner-side sampler (ref learner/data_loader.cpp), one slave thread."""

    def __init__(self, conf):
        self.L = lib()
        self.h = self.L.mzo_loader_create(conf.encode())
        assert self.h, "oracle loader create failed: " + conf

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mzo_loader_destroy(self.h)
            self.h = None

    def load_data_from_file(self, path): self.L.mzo_loader_load_file(self.h, path.encode())
    def num_data(self): return self.L.mzo_loader_num_data(self.h)
    def num_games(self): return self.L.mzo_loader_num_games(self.h)

    def sample_data(self, features, action_features, policy, value, reward, loss_scale, sampled_index):
        self.L.mzo_loader_sample(self.h, fptr(features), fptr(action_features), fptr(policy), fptr(value), fptr(reward), fptr(loss_scale), iptr(sampled_index))

    def update_priority(self, sampled_index, batch_values):
        self.L.mzo_loader_update_priority(self.h, iptr(np.ascontiguousarray(sampled_index, np.int32)), fptr(np.ascontiguousarray(batch_values, np.float32)))


class OracleGroup:
    def __init__(self, conf, desc, raw):
        self.L = lib()
        self.raw = np.ascontiguousarray(raw, np.float32)
        self.h = self.L.mzo_group_create(conf.encode(), C.byref(desc), fptr(self.raw), self.raw.size)
        assert self.h, "oracle group create failed: " + conf
        self._read = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mzo_group_destroy(self.h)
            self.h = None

    def set_trace(self, on=True):
        self.L.mzo_group_set_trace(self.h, int(on))

    def cycles(self, n):
        return self.L.mzo_group_cycles(self.h, n)

    def command(self, line, weights=None):
        """One stdin-protocol line between two cycles (ref actor_group.cpp:200-252); `weights` = the parameters of the file a load_model line names."""
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        rc = self.L.mzo_group_command(self.h, line.encode(), None if w is None else fptr(w), 0 if w is None else w.size)
        assert rc >= 0, "oracle refused: " + line
        return rc

    def num_cycles(self):
        return self.L.mzo_group_num_cycles(self.h)

    def leaf_evals(self):
        return self.L.mzo_group_leaf_evals(self.h)

    def games(self):
        return self.L.mzo_group_games(self.h)

    def _strings(self, count_fn, get_fn):
        out = []
        buf = C.create_string_buffer(1 << 20)
        for i in range(count_fn(self.h)):
            n = get_fn(self.h, i, buf, len(buf))
            if n >= len(buf):
                buf = C.create_string_buffer(n + 1)
                get_fn(self.h, i, buf, len(buf))
            out.append(buf.value.decode())
        return out

    def lines(self):
        return self._strings(self.L.mzo_group_num_lines, self.L.mzo_group_line)

    def trace(self):
        return self._strings(self.L.mzo_group_num_trace, self.L.mzo_group_trace)

    def peek_records(self, games):
        """Records of the games as they stand (unfinished ones included)."""
        return self._strings(lambda h: games, self.L.mzo_group_peek_record)


class OracleEnv:
    def __init__(self, conf):
        self.L = lib()
        self.h = self.L.mzo_env_create(conf.encode())
        assert self.h, "oracle env create failed: " + conf

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mzo_env_destroy(self.h)
            self.h = None

    def reset(self): self.L.mzo_env_reset(self.h)
    def act(self, a, player=None): return bool(self.L.mzo_env_act(self.h, a, self.turn() if player is None else player))
    def turn(self): return self.L.mzo_env_turn(self.h)
    def is_terminal(self): return bool(self.L.mzo_env_is_terminal(self.h))
    def eval_score(self, resign=False): return self.L.mzo_env_eval_score(self.h, int(resign))
    def policy_size(self): return self.L.mzo_env_policy_size(self.h)
    def reward(self): return self.L.mzo_env_reward(self.h)
    def seed(self): return self.L.mzo_env_seed(self.h)

    def legal_mask(self):
        m = np.zeros(self.policy_size(), np.uint8)
        self.L.mzo_env_legal_mask(self.h, m.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return m

    def features(self, rot=0):
        bs = self.L.mzo_env_board_size(self.h)
        f = np.empty(self.L.mzo_env_num_input_channels(self.h) * bs * bs, np.float32)
        n = self.L.mzo_env_features(self.h, rot, fptr(f))
        assert n == f.size
        return f


class OracleTree:
    def __init__(self, conf, tree_node_size):
        self.L = lib()
        self.h = self.L.mzo_tree_create(conf.encode(), tree_node_size)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mzo_tree_destroy(self.h)
            self.h = None

    def reset(self, root_player): self.L.mzo_tree_reset(self.h, root_player)

    def select(self, start=-1, cap=4096):
        p = np.empty(cap, np.int32)
        n = self.L.mzo_tree_select(self.h, start, iptr(p), cap)
        return p[:n].copy()

    def expand_backup(self, actions, player, policy, logit, value, reward=0.0):
        a = np.ascontiguousarray(actions, np.int32)
        p = np.ascontiguousarray(policy, np.float32)
        l = np.ascontiguousarray(logit, np.float32)
        self.L.mzo_tree_expand_backup(self.h, a.size, iptr(a), player, fptr(p), fptr(l), float(value), float(reward))

    def set_child_policy(self, node, policy, logit, noise):
        self.L.mzo_tree_set_child_policy(self.h, node, float(policy), float(logit), float(noise))

    def num_nodes(self): return self.L.mzo_tree_num_nodes(self.h)

    def dump(self):
        n = self.num_nodes()
        ints = [np.empty(n, np.int32) for _ in range(4)]
        fl = [np.empty(n, np.float32) for _ in range(7)]
        self.L.mzo_tree_dump(self.h, n, *[iptr(x) for x in ints], *[fptr(x) for x in fl])
        keys = ["action", "player", "num_children", "first_child", "mean", "count", "policy", "logit", "noise", "value", "reward"]
        return dict(zip(keys, ints + fl))
