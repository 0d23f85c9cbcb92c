"""ctypes binding of include/mzgpu.h (the C-ABI drop-in boundary)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MZ_HOST, MZ_DEVICE = 0, 1


class MzError(RuntimeError):
    pass


def lib_path():
    # MZ_LIBMZGPU: another build of the same library (tools/gpu_sanitize.sh: the host objects under AddressSanitizer / ThreadSanitizer); never a fallback
    return os.environ.get("MZ_LIBMZGPU") or os.path.join(_HERE, "libmzgpu.so")


class NetDesc(C.Structure):
    _fields_ = [("game_name", C.c_char * 64),
                ("num_input_channels", C.c_int), ("input_channel_height", C.c_int), ("input_channel_width", C.c_int),
                ("num_hidden_channels", C.c_int), ("hidden_channel_height", C.c_int), ("hidden_channel_width", C.c_int),
                ("num_action_feature_channels", C.c_int), ("num_blocks", C.c_int), ("action_size", C.c_int),
                ("num_value_hidden_channels", C.c_int), ("discrete_value_size", C.c_int), ("type", C.c_int)]


class SearchCfg(C.Structure):
    _fields_ = [("num_simulation", C.c_int), ("puct_base", C.c_float), ("puct_init", C.c_float), ("reward_discount", C.c_float),
                ("value_rescale", C.c_int), ("flipping_player", C.c_int), ("atari_init_q", C.c_int)]


class WorkerStats(C.Structure):
    _fields_ = [("cycles", C.c_uint64), ("leaf_evals", C.c_uint64), ("moves", C.c_uint64), ("games", C.c_uint64),
                ("ms_select", C.c_double), ("ms_env", C.c_double), ("ms_forward", C.c_double), ("ms_expand", C.c_double),
                ("ms_move", C.c_double), ("ms_total", C.c_double), ("sim_launches", C.c_uint64), ("sim_cycles", C.c_uint64),
                ("pre_evals", C.c_uint64), ("pre_hits", C.c_uint64), ("pre_alt_hits", C.c_uint64), ("pre_launches", C.c_uint64),
                ("pre_batch_launches", C.c_uint64), ("pre_pair_launches", C.c_uint64)]


NET_TYPES = {"alphazero": 0, "muzero": 1, "muzero_atari": 2}


def make_desc(game, cin, h, w, ch, hh, hw, ac, blocks, actions, vh=256, dv=1, type_name="alphazero"):
    """Same argument order as the reference's create_network (network/py/create_network.py:6-18)."""
    d = NetDesc()
    d.game_name = game.encode()
    (d.num_input_channels, d.input_channel_height, d.input_channel_width) = (cin, h, w)
    (d.num_hidden_channels, d.hidden_channel_height, d.hidden_channel_width) = (ch, hh, hw)
    (d.num_action_feature_channels, d.num_blocks, d.action_size) = (ac, blocks, actions)
    (d.num_value_hidden_channels, d.discrete_value_size, d.type) = (vh, dv, NET_TYPES[type_name])
    return d


# BASELINE.json configs: network shapes and search configuration strings
DESCS = {
    "c1": lambda: make_desc("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9),
    "c2": lambda: make_desc("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82),
    "c3": lambda: make_desc("othello_8x8", 4, 8, 8, 64, 8, 8, 1, 6, 65),
    "c4": lambda: make_desc("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, type_name="muzero"),
    "c5": lambda: make_desc("atari_ms_pacman", 32, 96, 96, 64, 6, 6, 18, 6, 18, 256, 601, type_name="muzero_atari"),
}
CONFIGS = {
    "c1": "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=8:zero_num_threads=1",
    "c2": "env_game=go:env_board_size=9:actor_num_simulation=400:zero_num_parallel_games=256",
    "c3": ("env_game=othello:env_board_size=8:actor_num_simulation=16:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
           "actor_use_gumbel_noise=true:actor_gumbel_sample_size=16:actor_gumbel_sigma_visit_c=50:actor_gumbel_sigma_scale_c=1:"
           "zero_num_parallel_games=1024"),
    "c4": "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=50:zero_num_parallel_games=256",
    # per GPU (512 games over 8 GPUs); synthetic Atari-shaped environment (ALE / ROMs are not available)
    "c5": ("env_game=atari:nn_type_name=muzero:actor_num_simulation=50:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
           "actor_use_gumbel_noise=true:actor_gumbel_sample_size=16:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:"
           "actor_mcts_reward_discount=0.997:atari_init_q=true:zero_actor_intermediate_sequence_length=200:learner_n_step_return=10:"
           "zero_num_parallel_games=64"),
}


def load():
    """Load libmzgpu.so (fail loudly: the product has no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise MzError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
    L = C.CDLL(path)
    fp, ip, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.mz_last_error.restype = C.c_char_p
    L.mz_net_param_count.restype = C.c_long
    L.mz_net_param_count.argtypes = [C.POINTER(NetDesc)]
    L.mz_net_generate_weights.argtypes = [C.POINTER(NetDesc), C.c_uint64, fp]
    L.mz_net_read_pt.argtypes = [C.c_char_p, C.POINTER(NetDesc), fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mz_net_create.restype = vp
    L.mz_net_create.argtypes = [C.c_int, C.POINTER(NetDesc), fp, C.c_size_t]
    L.mz_net_reload.argtypes = [vp, fp, C.c_size_t]
    L.mz_net_destroy.argtypes = [vp]
    L.mz_net_get_desc.argtypes = [vp, C.POINTER(NetDesc)]
    L.mz_net_set_precision.argtypes = [vp, C.c_int]
    L.mz_net_forward_az.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int]
    L.mz_net_initial.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int]
    L.mz_net_recurrent.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    L.mz_net_time_forward.argtypes = [vp, C.c_int, C.c_int, fp, fp, C.POINTER(C.c_double)]
    L.mz_net_time_tower_conv.argtypes = [vp, C.c_int, C.c_int, fp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mz_pool_create.restype = vp
    L.mz_pool_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SearchCfg)]
    L.mz_pool_destroy.argtypes = [vp]
    L.mz_pool_reset_search.argtypes = [vp, u8p, ip]
    L.mz_pool_select.argtypes = [vp, ip, ip, ip, ip]
    L.mz_pool_max_depth.argtypes = [vp]
    L.mz_pool_expand_backup.argtypes = [vp, ip, ip, fp, fp, ip, fp, fp]
    L.mz_pool_root_set_noise.argtypes = [vp, u8p, fp, fp, fp]
    L.mz_pool_root_read.argtypes = [vp, ip, ip] + [fp] * 12 + [ip]
    L.mz_pool_read_nodes.argtypes = [vp, C.c_int, C.c_int, ip, ip, ip, ip] + [fp] * 7
    L.mz_pool_num_nodes.argtypes = [vp, C.c_int]
    if hasattr(L, "mz_worker_create"):
        L.mz_worker_create.restype = vp
        L.mz_worker_create.argtypes = [C.c_int, C.c_char_p, C.POINTER(NetDesc), fp, C.c_size_t]
        L.mz_worker_destroy.argtypes = [vp]
        L.mz_worker_command.argtypes = [vp, C.c_char_p]
        L.mz_worker_set_weights.argtypes = [vp, fp, C.c_size_t]
        L.mz_worker_load_model.argtypes = [vp, C.c_char_p, C.POINTER(NetDesc), fp, C.c_size_t]
        L.mz_weights_read.restype = vp
        L.mz_weights_read.argtypes = [C.c_char_p]
        L.mz_weights_desc.restype = C.POINTER(NetDesc)
        L.mz_weights_desc.argtypes = [vp]
        L.mz_weights_data.restype = fp
        L.mz_weights_data.argtypes = [vp]
        L.mz_weights_count.restype = C.c_size_t
        L.mz_weights_count.argtypes = [vp]
        L.mz_weights_free.argtypes = [vp]
        L.mz_weight_file_reads.restype = C.c_uint64
        L.mz_worker_run_cycles.argtypes = [vp, C.c_int]
        L.mz_worker_cycles_per_move.argtypes = [vp]
        L.mz_worker_lanes.argtypes = [vp]
        L.mz_net_read_weight_file.argtypes = [C.c_char_p, C.POINTER(NetDesc), fp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mz_worker_pop_line.argtypes = [vp, C.c_char_p, C.c_int]
        L.mz_worker_wait_lines.argtypes = [vp]
        L.mz_worker_get_stats.argtypes = [vp, C.POINTER(WorkerStats)]
        L.mz_worker_peek_record.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
        L.mz_worker_net.restype = vp
        L.mz_worker_net.argtypes = [vp]
        L.mz_env_create.restype = vp
        L.mz_env_create.argtypes = [C.c_char_p]
        L.mz_env_destroy.argtypes = [vp]
        L.mz_env_reset.argtypes = [vp]
        L.mz_env_reset_seed.argtypes = [vp, C.c_int]
        L.mz_env_reward.restype = C.c_float
        L.mz_env_reward.argtypes = [vp]
        L.mz_env_act.argtypes = [vp, C.c_int, C.c_int]
        for n in ("mz_env_turn", "mz_env_is_terminal", "mz_env_policy_size", "mz_env_feature_size"):
            getattr(L, n).argtypes = [vp]
        L.mz_env_eval_score.restype = C.c_float
        L.mz_env_eval_score.argtypes = [vp, C.c_int]
        L.mz_env_legal_mask.argtypes = [vp, u8p]
        L.mz_env_features.argtypes = [vp, C.c_int, fp]
        L.mz_env_feature_bits.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32)]
        L.mz_env_action_from_string.argtypes = [vp, C.c_char_p]
        L.mz_worker_create_shared.restype = vp
        L.mz_worker_create_shared.argtypes = [C.c_int, C.c_char_p, vp]
        L.mz_godev_playout.argtypes = [C.c_int, C.c_int, C.c_float, ip, C.c_int, C.c_int, ip, C.POINTER(C.c_uint32), u8p, ip, fp, ip]
        L.mz_envdev_playout.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_float, ip, C.c_int, C.c_int, ip, C.POINTER(C.c_uint32), u8p, ip, fp, ip]
        L.mz_sort_candidates.argtypes = [C.c_int, fp, C.c_int, ip]
        L.mz_invert_values_device.argtypes = [C.c_int, fp, C.c_int, fp]
    L.mz_loader_create.restype = vp
    L.mz_loader_create.argtypes = [C.c_int, C.c_char_p]
    L.mz_loader_destroy.argtypes = [vp]
    L.mz_loader_load_data_from_file.argtypes = [vp, C.c_char_p]
    L.mz_loader_add_record.argtypes = [vp, C.c_char_p]
    L.mz_loader_sample_data.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.mz_loader_update_priority.argtypes = [vp, vp, vp]
    for n in ("mz_loader_num_data", "mz_loader_num_games"):
        getattr(L, n).argtypes = [vp]
    L.mz_loader_shape.argtypes = [vp, C.c_int]
    for n in ("mz_invert_value", "mz_transform_value"):
        getattr(L, n).restype = C.c_float
        getattr(L, n).argtypes = [C.c_float]
    _LIB = L
    return L


def _err(L):
    return (L.mz_last_error() or b"").decode(errors="replace")


def _check(L, rc):
    if rc < 0:
        raise MzError(f"libmzgpu error {rc}: {_err(L)}")
    return rc


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def device_count():
    return load().mz_device_count()


def usable_cpus():
    return load().mz_usable_cpus()


def read_pt(path):
    """(desc, weights) of a TorchScript file written by the reference's trainer — native parser, no torch."""
    L = load()
    d = NetDesc()
    n = C.c_size_t(0)
    _check(L, L.mz_net_read_pt(path.encode(), C.byref(d), None, 0, C.byref(n)))
    w = np.empty(n.value, np.float32)
    _check(L, L.mz_net_read_pt(path.encode(), C.byref(d), _f(w), n.value, C.byref(n)))
    return d, w


def read_weight_file(path):
    """(desc, weights) of `path` as Network::loadModel is given it: the TorchScript archive, or for a missing x.pt its sibling x.mzw."""
    L = load()
    d = NetDesc()
    n = C.c_size_t(0)
    _check(L, L.mz_net_read_weight_file(path.encode(), C.byref(d), None, 0, C.byref(n)))
    w = np.empty(n.value, np.float32)
    _check(L, L.mz_net_read_weight_file(path.encode(), C.byref(d), _f(w), n.value, C.byref(n)))
    return d, w


def weight_file_reads():
    """weight files this process has opened and parsed through libmzgpu so far"""
    return int(load().mz_weight_file_reads())


def read_weights_once(path):
    """mz_weights_read: (desc, blob) from ONE open + parse of `path` (.pt, or its .mzw sibling)"""
    L = load()
    h = L.mz_weights_read(path.encode())
    if not h:
        raise MzError("mz_weights_read failed: " + _err(L))
    try:
        d = NetDesc.from_buffer_copy(L.mz_weights_desc(h).contents)
        n = L.mz_weights_count(h)
        w = np.ctypeslib.as_array(L.mz_weights_data(h), shape=(n,)).copy()
    finally:
        L.mz_weights_free(h)
    return d, w


def param_count(desc):
    return _check(load(), load().mz_net_param_count(C.byref(desc)))


def generate_weights(desc, seed=0):
    L = load()
    w = np.empty(param_count(desc), np.float32)
    _check(L, L.mz_net_generate_weights(C.byref(desc), seed, _f(w)))
    return w


class Net:
    """Mirror of minizero::network::{AlphaZeroNetwork,MuZeroNetwork} (ref network/alphazero_network.h, muzero_network.h)."""

    def __init__(self, desc, weights, device=0, handle=None):
        self.L = load()
        self.desc = desc
        self._own = handle is None
        if handle is None:
            w = np.ascontiguousarray(weights, np.float32)
            handle = self.L.mz_net_create(device, C.byref(desc), _f(w), w.size)
            if not handle:
                raise MzError("mz_net_create failed: " + _err(self.L))
        self.h = handle

    def close(self):
        if getattr(self, "h", None) and self._own:
            self.L.mz_net_destroy(self.h)
        self.h = None

    __del__ = close

    def reload(self, weights):
        w = np.ascontiguousarray(weights, np.float32)
        _check(self.L, self.L.mz_net_reload(self.h, _f(w), w.size))

    def set_precision(self, name):
        """'f32' (default) or 'bf16x3' (opt-in split-bf16 tower, outputs within 1e-3 of f32)"""
        _check(self.L, self.L.mz_net_set_precision(self.h, {"f32": 0, "bf16x3": 1}[name]))

    def hidden_size(self):
        d = self.desc
        return d.num_hidden_channels * d.hidden_channel_height * d.hidden_channel_width

    def forward(self, features):
        x = np.ascontiguousarray(features, np.float32)
        B, A = x.shape[0], self.desc.action_size
        p, l, v = np.empty((B, A), np.float32), np.empty((B, A), np.float32), np.empty(B, np.float32)
        _check(self.L, self.L.mz_net_forward_az(self.h, x.ctypes.data, B, p.ctypes.data, l.ctypes.data, v.ctypes.data, MZ_HOST))
        return p, l, v

    def initial_inference(self, features):
        x = np.ascontiguousarray(features, np.float32)
        B, A = x.shape[0], self.desc.action_size
        p, l, v = np.empty((B, A), np.float32), np.empty((B, A), np.float32), np.empty(B, np.float32)
        h = np.empty((B, self.hidden_size()), np.float32)
        _check(self.L, self.L.mz_net_initial(self.h, x.ctypes.data, B, p.ctypes.data, l.ctypes.data, v.ctypes.data, h.ctypes.data, MZ_HOST))
        return p, l, v, h

    def recurrent_inference(self, hidden, action_plane):
        hin = np.ascontiguousarray(hidden, np.float32)
        act = np.ascontiguousarray(action_plane, np.float32)
        B, A = hin.shape[0], self.desc.action_size
        p, l, v = np.empty((B, A), np.float32), np.empty((B, A), np.float32), np.empty(B, np.float32)
        r, h = np.empty(B, np.float32), np.empty((B, self.hidden_size()), np.float32)
        _check(self.L, self.L.mz_net_recurrent(self.h, hin.ctypes.data, act.ctypes.data, B, p.ctypes.data, l.ctypes.data, v.ctypes.data,
                                               r.ctypes.data, h.ctypes.data, MZ_HOST))
        return p, l, v, r, h

    def time_forward(self, batch, iters):
        t, c, fl = C.c_float(), C.c_float(), C.c_double()
        _check(self.L, self.L.mz_net_time_forward(self.h, batch, iters, C.byref(t), C.byref(c), C.byref(fl)))
        return t.value, c.value, fl.value


    def time_tower_conv(self, batch, iters):
        t, fl, by = C.c_float(), C.c_double(), C.c_double()
        _check(self.L, self.L.mz_net_time_tower_conv(self.h, batch, iters, C.byref(t), C.byref(fl), C.byref(by)))
        return t.value, fl.value, by.value


class Pool:
    """Mirror of minizero::actor::MCTS for `games` trees at once (ref actor/mcts.h:75-119)."""

    def __init__(self, games, nodes_per_game, action_size, num_simulation, puct_base=19652.0, puct_init=1.25, reward_discount=1.0,
                 value_rescale=False, flipping_player=2, atari_init_q=False, device=0):
        self.L = load()
        self.games, self.A = games, action_size
        cfg = SearchCfg(num_simulation, puct_base, puct_init, reward_discount, int(value_rescale), flipping_player, int(atari_init_q))
        self.h = self.L.mz_pool_create(device, games, nodes_per_game, action_size, C.byref(cfg))
        if not self.h:
            raise MzError("mz_pool_create failed: " + _err(self.L))
        self.max_depth = self.L.mz_pool_max_depth(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.mz_pool_destroy(self.h)
        self.h = None

    __del__ = close

    def reset_search(self, root_player, mask=None):
        rp = np.ascontiguousarray(root_player, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
        _check(self.L, self.L.mz_pool_reset_search(self.h, m, _i(rp)))

    def select(self, start_node=None):
        G = self.games
        pl = np.empty(G, np.int32)
        paths = np.empty((G, self.max_depth), np.int32)
        acts = np.empty((G, self.max_depth), np.int32)
        st = None if start_node is None else _i(np.ascontiguousarray(start_node, np.int32))
        _check(self.L, self.L.mz_pool_select(self.h, st, _i(pl), _i(paths), _i(acts)))
        return pl, paths, acts

    def expand_backup(self, cand_count, cand_action, cand_policy, cand_logit, cand_player, value, reward=None):
        cc = np.ascontiguousarray(cand_count, np.int32)
        ca = np.ascontiguousarray(cand_action, np.int32)
        cp = np.ascontiguousarray(cand_policy, np.float32)
        cl = np.ascontiguousarray(cand_logit, np.float32)
        pl = np.ascontiguousarray(cand_player, np.int32)
        v = np.ascontiguousarray(value, np.float32)
        r = np.zeros(self.games, np.float32) if reward is None else np.ascontiguousarray(reward, np.float32)
        assert ca.size == self.games * self.A
        _check(self.L, self.L.mz_pool_expand_backup(self.h, _i(cc), _i(ca), _f(cp), _f(cl), _i(pl), _f(v), _f(r)))

    def root_set_noise(self, policy, logit, noise, mask=None):
        p = np.ascontiguousarray(policy, np.float32)
        l = np.ascontiguousarray(logit, np.float32)
        n = np.ascontiguousarray(noise, np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
        _check(self.L, self.L.mz_pool_root_set_noise(self.h, m, _f(p), _f(l), _f(n)))

    def root_read(self):
        G, A = self.games, self.A
        nc, act, bs = np.empty(G, np.int32), np.empty((G, A), np.int32), np.empty(G, np.int32)
        fa = [np.empty((G, A), np.float32) for _ in range(7)]
        fg = [np.empty(G, np.float32) for _ in range(5)]
        _check(self.L, self.L.mz_pool_root_read(self.h, _i(nc), _i(act), *[_f(x) for x in fa], *[_f(x) for x in fg], _i(bs)))
        keys = ["count", "mean", "policy", "logit", "noise", "value", "reward"]
        out = dict(zip(keys, fa))
        out.update(num_children=nc, action=act, bound_size=bs, root_count=fg[0], root_mean=fg[1], root_value=fg[2], bound_lo=fg[3], bound_hi=fg[4])
        return out

    def num_nodes(self, game):
        return _check(self.L, self.L.mz_pool_num_nodes(self.h, game))

    def read_nodes(self, game, n=None):
        n = self.num_nodes(game) if n is None else n
        ints = [np.empty(n, np.int32) for _ in range(4)]
        fl = [np.empty(n, np.float32) for _ in range(7)]
        _check(self.L, self.L.mz_pool_read_nodes(self.h, game, n, *[_i(x) for x in ints], *[_f(x) for x in fl]))
        keys = ["action", "player", "num_children", "first_child", "mean", "count", "policy", "logit", "noise", "value", "reward"]
        return dict(zip(keys, ints + fl))


class Worker:
    """Mirror of minizero::actor::ActorGroup (`-mode sp`, ref actor/actor_group.cpp:136-252)."""

    def __init__(self, conf, desc=None, weights=None, device=0, shared=None):
        """desc / weights None: the network is read from the configuration's nn_file_name (.pt, or its .mzw sibling);
        shared: a Net the worker runs ON instead of loading its own copy (BaseActor::setNetwork, mz_worker_create_shared)"""
        self.L = load()
        self.shared = shared  # keeps the caller's network alive
        if shared is not None:
            self.h = self.L.mz_worker_create_shared(device, conf.encode(), shared.h)
            desc = shared.desc
        elif desc is None:
            self.h = self.L.mz_worker_create(device, conf.encode(), None, None, 0)
        else:
            w = np.ascontiguousarray(weights, np.float32)
            self.h = self.L.mz_worker_create(device, conf.encode(), C.byref(desc), _f(w), w.size)
        if not self.h:
            raise MzError("mz_worker_create failed: " + _err(self.L))
        if desc is None:
            desc = NetDesc()
            _check(self.L, self.L.mz_net_get_desc(self.L.mz_worker_net(self.h), C.byref(desc)))
        self.desc = desc

    def close(self):
        if getattr(self, "h", None):
            self.L.mz_worker_destroy(self.h)
        self.h = None

    __del__ = close

    def command(self, line):
        return _check(self.L, self.L.mz_worker_command(self.h, line.encode()))

    def set_weights(self, weights):
        w = np.ascontiguousarray(weights, np.float32)
        _check(self.L, self.L.mz_worker_set_weights(self.h, _f(w), w.size))

    def load_model(self, path, desc, weights):
        """`load_model <path>` for a caller that has read the file already (one read for all devices: mz_weights_read)"""
        w = np.ascontiguousarray(weights, np.float32)
        return _check(self.L, self.L.mz_worker_load_model(self.h, path.encode(), C.byref(desc), _f(w), w.size))

    def run_cycles(self, n):
        return _check(self.L, self.L.mz_worker_run_cycles(self.h, n))

    def cycles_per_move(self):
        return _check(self.L, self.L.mz_worker_cycles_per_move(self.h))

    def lanes(self):
        return _check(self.L, self.L.mz_worker_lanes(self.h))

    def pop_lines(self, wait=True):
        """Every finished record.  wait=False: only those that are complete right now (mz_worker_pop_line never blocks: an Atari record whose OBS tag
        is still being compressed, and everything behind it, stays queued) — what a driver does between two moves."""
        if wait:
            _check(self.L, self.L.mz_worker_wait_lines(self.h))
        out = []
        buf = C.create_string_buffer(1 << 20)
        while True:
            need = _check(self.L, self.L.mz_worker_pop_line(self.h, None, 0))  # size query: Atari records carry megabytes of OBS hex
            if need == 0:
                break
            if need >= len(buf):
                buf = C.create_string_buffer(need + 1)
            _check(self.L, self.L.mz_worker_pop_line(self.h, buf, len(buf)))
            out.append(buf.value.decode())
        return out

    def peek_records(self, games):
        """Records of games 0..games-1 as they stand (unfinished ones included)."""
        buf = C.create_string_buffer(1 << 20)
        out = []
        for g in range(games):
            need = _check(self.L, self.L.mz_worker_peek_record(self.h, g, None, 0))
            if need >= len(buf):
                buf = C.create_string_buffer(need + 1)
            _check(self.L, self.L.mz_worker_peek_record(self.h, g, buf, len(buf)))
            out.append(buf.value.decode())
        return out

    def stats(self):
        s = WorkerStats()
        _check(self.L, self.L.mz_worker_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in WorkerStats._fields_}

    def net(self):
        return Net(self.desc, None, handle=self.L.mz_worker_net(self.h))


class DataLoader:
    """Mirror of the reference's learner-side `minizero_py.DataLoader` (ref learner/pybind.cpp:62-84, learner/data_loader.h:75-93): same
    method names and argument order.  `conf` is the configuration STRING (the reference takes a .cfg file name; pass its lines joined with
    ':' plus env_game).  Arrays may be numpy arrays (host) or torch CUDA tensors (the batch is then written in place on the device)."""

    def __init__(self, conf, device=0):
        self.L = load()
        self.h = self.L.mz_loader_create(device, conf.encode())
        if not self.h:
            raise MzError("mz_loader_create failed: " + _err(self.L))

    def close(self):
        if getattr(self, "h", None):
            self.L.mz_loader_destroy(self.h)
        self.h = None

    __del__ = close

    def initialize(self):  # ref data_loader.cpp:194-198: thread creation; nothing to do here
        pass

    def load_data_from_file(self, file_name):
        return _check(self.L, self.L.mz_loader_load_data_from_file(self.h, file_name.encode()))

    def add_record(self, line):
        return _check(self.L, self.L.mz_loader_add_record(self.h, line.encode()))

    def num_data(self): return _check(self.L, self.L.mz_loader_num_data(self.h))
    def num_games(self): return _check(self.L, self.L.mz_loader_num_games(self.h))

    def shapes(self):
        """(batch, floats per sample of features, action_features, policy, value, reward)"""
        return tuple(_check(self.L, self.L.mz_loader_shape(self.h, k)) for k in range(6))

    @staticmethod
    def _ptr(a):
        if a is None:
            return None, None
        if hasattr(a, "data_ptr"):  # torch tensor
            assert a.is_contiguous()
            return a.data_ptr(), bool(a.is_cuda)
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data, False

    def sample_data(self, features, action_features, policy, value, reward, loss_scale, sampled_index):
        ptrs = [self._ptr(a) for a in (features, action_features, policy, value, reward, loss_scale, sampled_index)]
        sides = {d for p, d in ptrs if p is not None}
        if len(sides) != 1:
            raise MzError("sample_data: all buffers must be on the same side (host numpy arrays or CUDA tensors)")
        _check(self.L, self.L.mz_loader_sample_data(self.h, *[p for p, _ in ptrs], MZ_DEVICE if sides.pop() else MZ_HOST))

    def update_priority(self, sampled_index, batch_values):
        si = np.ascontiguousarray(sampled_index, np.int32)
        bv = np.ascontiguousarray(batch_values, np.float32)
        _check(self.L, self.L.mz_loader_update_priority(self.h, si.ctypes.data, bv.ctypes.data))


class Env:
    """Host rules engine used by the worker for AlphaZero leaves (mirror of BaseEnv, ref environment/base/base_env.h:74-114)."""

    def __init__(self, conf):
        self.L = load()
        self.h = self.L.mz_env_create(conf.encode())
        if not self.h:
            raise MzError("mz_env_create failed: " + _err(self.L))

    def close(self):
        if getattr(self, "h", None):
            self.L.mz_env_destroy(self.h)
        self.h = None

    __del__ = close

    def reset(self): self.L.mz_env_reset(self.h)
    def reset_seed(self, seed): self.L.mz_env_reset_seed(self.h, seed)
    def reward(self): return self.L.mz_env_reward(self.h)
    def act(self, a, player=None): return bool(self.L.mz_env_act(self.h, a, self.turn() if player is None else player))
    def turn(self): return self.L.mz_env_turn(self.h)
    def is_terminal(self): return bool(self.L.mz_env_is_terminal(self.h))
    def eval_score(self, resign=False): return self.L.mz_env_eval_score(self.h, int(resign))
    def policy_size(self): return self.L.mz_env_policy_size(self.h)
    def action_from_string(self, s): return self.L.mz_env_action_from_string(self.h, s.encode())

    def legal_mask(self):
        m = np.zeros(self.policy_size(), np.uint8)
        self.L.mz_env_legal_mask(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8)))
        return m

    def features(self, rot=0):
        f = np.empty(self.L.mz_env_feature_size(self.h), np.float32)
        self.L.mz_env_features(self.h, rot, _f(f))
        return f

    def feature_bits(self, rot, channels, points):
        w = np.zeros(channels * ((points + 31) // 32), np.uint32)
        _check(self.L, self.L.mz_env_feature_bits(self.h, rot, w.ctypes.data_as(C.POINTER(C.c_uint32))))
        return w


def godev_playout(board_size, komi, actions, root_prefix, rots, device=0):
    """Device Go engine: root = actions[:root_prefix] on the host engine, then one device move per remaining action.
    Returns (feat_bits [steps][18*W32], legal [steps][A], terminal [steps], eval [steps], player [steps])."""
    L = load()
    P = board_size * board_size
    acts = np.ascontiguousarray(actions, np.int32)
    steps = len(acts) - root_prefix + 1
    rots = np.ascontiguousarray(rots, np.int32)
    assert len(rots) >= steps
    W32 = (P + 31) // 32
    feat = np.zeros((steps, 18 * W32), np.uint32)
    legal = np.zeros((steps, P + 1), np.uint8)
    term = np.zeros(steps, np.int32)
    ev = np.zeros(steps, np.float32)
    pl = np.zeros(steps, np.int32)
    _check(L, L.mz_godev_playout(device, board_size, komi, _i(acts), len(acts), root_prefix, _i(rots), feat.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 legal.ctypes.data_as(C.POINTER(C.c_uint8)), _i(term), _f(ev), _i(pl)))
    return feat, legal, term, ev, pl


def envdev_playout(game, board_size, komi, actions, root_prefix, rots, channels, num_actions, device=0):
    """Device rules engine of `game` ("go", "othello", "tictactoe"): root = actions[:root_prefix] on the host engine, then one device move per
    remaining action.  Returns (feat_bits [steps][channels*W32], legal [steps][num_actions], terminal, eval, player)."""
    L = load()
    P = board_size * board_size
    acts = np.ascontiguousarray(actions, np.int32)
    steps = len(acts) - root_prefix + 1
    rots = np.ascontiguousarray(rots, np.int32)
    assert len(rots) >= steps
    W32 = (P + 31) // 32
    feat = np.zeros((steps, channels * W32), np.uint32)
    legal = np.zeros((steps, num_actions), np.uint8)
    term = np.zeros(steps, np.int32)
    ev = np.zeros(steps, np.float32)
    pl = np.zeros(steps, np.int32)
    _check(L, L.mz_envdev_playout(device, game.encode(), board_size, komi, _i(acts), len(acts), root_prefix, _i(rots),
                                  feat.ctypes.data_as(C.POINTER(C.c_uint32)), legal.ctypes.data_as(C.POINTER(C.c_uint8)), _i(term), _f(ev), _i(pl)))
    return feat, legal, term, ev, pl


def invert_values_device(values, device=0):
    """invertValue (601-bin decode) of every element by the device function of the simulation kernel."""
    L = load()
    v = np.ascontiguousarray(values, np.float32)
    out = np.zeros(len(v), np.float32)
    _check(L, L.mz_invert_values_device(device, _f(v), len(v), _f(out)))
    return out


def sort_candidates(policy, device=0):
    """Order of candidates under the reference's std::sort(policy descending), computed by the device path."""
    L = load()
    p = np.ascontiguousarray(policy, np.float32)
    order = np.zeros(len(p), np.int32)
    _check(L, L.mz_sort_candidates(device, _f(p), len(p), _i(order)))
    return order
