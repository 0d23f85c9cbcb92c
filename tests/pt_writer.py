"""Writes a TorchScript `weight_iter_N.pt` a MiniZero worker can load — WITHOUT the reference's Python (which cannot travel to the GPU box).

The file format is fixed by the reference's trainer (`learner/train.py:127`: `torch.jit.script(network).save(path)`) and by what its
C++ side reads back (`network/network.cpp:14-42`: the `get_*` methods; the parameters by attribute name).  What a reader sees of such a file
is (a) the top-level class name, (b) the twelve hyper-parameter attributes and their exported getters, (c) the tree of sub-module / parameter
NAMES in registration order (`alphazero_network.py:21-40`, `muzero_network.py:7-84`, `muzero_atari_network.py:7-104`, `network_unit.py:6-78`).
This module rebuilds exactly that from the `_tree_*` tables — the layer tree as data, generic containers instead of one class per unit —
and fills it with a flat parameter blob in the worker's order (`mz_net_create`).  Its forward is the identity: the tests use the file as a
weight container for `load_model`, never as a network (the network math is pinned by tests/golden/nn_*.npz).

tests/test_pt_reader.py::test_own_writer_has_the_reference_layout checks, where /root/reference exists, that the state_dict keys, shapes and
getter values equal those of the reference's own `create_network(...)` module for every network type.
"""
import numpy as np
import torch
import torch.nn as nn


class _Bag(nn.Module):
    """An ordered bag of named children (a unit of the network: stem, block, head)."""

    def __init__(self, children):
        super().__init__()
        for name, child in children:
            self.add_module(name, child)


def _conv(cin, cout, k=3, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2)


def _block(c):
    return _Bag([("conv1", _conv(c, c)), ("bn1", nn.BatchNorm2d(c)), ("conv2", _conv(c, c)), ("bn2", nn.BatchNorm2d(c))])


def _blocks(c, n):
    return nn.ModuleList([_block(c) for _ in range(n)])


def _policy_head(c, h, w, actions):
    oc = (actions + h * w - 1) // (h * w)  # network_unit.py:29-31: enough 1x1 planes to hold one value per action
    return _Bag([("conv", _conv(c, oc, 1)), ("bn", nn.BatchNorm2d(oc)), ("fc", nn.Linear(oc * h * w, actions))])


def _scalar_value_head(c, h, w, hidden):
    return _Bag([("conv", _conv(c, 1, 1)), ("bn", nn.BatchNorm2d(1)), ("fc1", nn.Linear(h * w, hidden)), ("fc2", nn.Linear(hidden, 1)), ("tanh", nn.Tanh())])


def _binned_value_head(c, h, w, hidden, bins):
    planes = (bins + h * w - 1) // (h * w)  # network_unit.py:73: enough 1x1 planes to hold one value per bin
    return _Bag([("conv", _conv(c, planes, 1)), ("bn", nn.BatchNorm2d(planes)), ("fc1", nn.Linear(h * w * planes, hidden)), ("fc2", nn.Linear(hidden, bins))])


def _value_head(d):
    if d["discrete_value_size"] == 1:
        return _scalar_value_head(d["num_hidden_channels"], d["hidden_channel_height"], d["hidden_channel_width"], d["num_value_hidden_channels"])
    return _binned_value_head(d["num_hidden_channels"], d["hidden_channel_height"], d["hidden_channel_width"], d["num_value_hidden_channels"], d["discrete_value_size"])


def _policy(d):
    return _policy_head(d["num_hidden_channels"], d["hidden_channel_height"], d["hidden_channel_width"], d["action_size"])


def _stem(cin, c, n):
    return _Bag([("conv", _conv(cin, c)), ("bn", nn.BatchNorm2d(c)), ("residual_blocks", _blocks(c, n))])


def _tree_alphazero(d):
    c = d["num_hidden_channels"]
    return [("conv", _conv(d["num_input_channels"], c)), ("bn", nn.BatchNorm2d(c)), ("residual_blocks", _blocks(c, d["num_blocks"])),
            ("policy", _policy(d)), ("value", _value_head(d))]


def _tree_muzero(d):
    c, n = d["num_hidden_channels"], d["num_blocks"]
    return [("representation_network", _stem(d["num_input_channels"], c, n)),
            ("dynamics_network", _stem(c + d["num_action_feature_channels"], c, n)),
            ("prediction_network", _Bag([("policy", _policy(d)), ("value", _value_head(d))]))]


def _tree_muzero_atari(d):
    c, n, h, w = d["num_hidden_channels"], d["num_blocks"], d["hidden_channel_height"], d["hidden_channel_width"]
    pool = lambda: nn.AvgPool2d(kernel_size=3, stride=2, padding=1)  # noqa: E731
    rep = _Bag([("conv1", _conv(d["num_input_channels"], c // 2, 3, 2)), ("bn1", nn.BatchNorm2d(c // 2)), ("residual_blocks1", _blocks(c // 2, 1)),
                ("conv2", _conv(c // 2, c, 3, 2)), ("bn2", nn.BatchNorm2d(c)), ("residual_blocks2", _blocks(c, 1)), ("avg_pooling1", pool()),
                ("residual_blocks3", _blocks(c, 1)), ("avg_pooling2", pool()), ("residual_blocks", _blocks(c, n))])
    dyn = _Bag([("conv", _conv(c + d["num_action_feature_channels"], c)), ("bn", nn.BatchNorm2d(c)), ("residual_blocks", _blocks(c, n)),
                ("reward_network", _binned_value_head(c, h, w, c, d["discrete_value_size"]))])
    return [("representation_network", rep), ("dynamics_network", dyn), ("prediction_network", _Bag([("policy", _policy(d)), ("value", _value_head(d))]))]


class _Hyper(nn.Module):
    """The hyper-parameter attributes and the exported getters the reference's C++ loader calls (network.cpp:27-41)."""

    def __init__(self, d, tree):
        super().__init__()
        self.game_name: str = d["game_name"]
        self.num_input_channels: int = d["num_input_channels"]
        self.input_channel_height: int = d["input_channel_height"]
        self.input_channel_width: int = d["input_channel_width"]
        self.num_hidden_channels: int = d["num_hidden_channels"]
        self.hidden_channel_height: int = d["hidden_channel_height"]
        self.hidden_channel_width: int = d["hidden_channel_width"]
        self.num_blocks: int = d["num_blocks"]
        self.action_size: int = d["action_size"]
        self.num_value_hidden_channels: int = d["num_value_hidden_channels"]
        self.discrete_value_size: int = d["discrete_value_size"]
        self._grow(d, tree)

    def _grow(self, d, tree):
        for name, child in tree:
            self.add_module(name, child)

    @torch.jit.export
    def get_game_name(self) -> str: return self.game_name
    @torch.jit.export
    def get_num_input_channels(self) -> int: return self.num_input_channels
    @torch.jit.export
    def get_input_channel_height(self) -> int: return self.input_channel_height
    @torch.jit.export
    def get_input_channel_width(self) -> int: return self.input_channel_width
    @torch.jit.export
    def get_num_hidden_channels(self) -> int: return self.num_hidden_channels
    @torch.jit.export
    def get_hidden_channel_height(self) -> int: return self.hidden_channel_height
    @torch.jit.export
    def get_hidden_channel_width(self) -> int: return self.hidden_channel_width
    @torch.jit.export
    def get_num_blocks(self) -> int: return self.num_blocks
    @torch.jit.export
    def get_action_size(self) -> int: return self.action_size
    @torch.jit.export
    def get_num_value_hidden_channels(self) -> int: return self.num_value_hidden_channels
    @torch.jit.export
    def get_discrete_value_size(self) -> int: return self.discrete_value_size

    def forward(self, state: torch.Tensor) -> torch.Tensor:
        return state  # a weight container (module docstring)


# the class NAMES are part of the format: the worker tells the three network types apart by them (ptfile.cpp), LibTorch by get_type_name()
class AlphaZeroNetwork(_Hyper):
    @torch.jit.export
    def get_type_name(self) -> str: return "alphazero"


class _HyperMz(_Hyper):
    """MuZero networks carry one more hyper-parameter, registered before the sub-networks (muzero_network.py:74)"""

    def _grow(self, d, tree):
        self.num_action_feature_channels: int = d["num_action_feature_channels"]
        super()._grow(d, tree)

    @torch.jit.export
    def get_num_action_feature_channels(self) -> int: return self.num_action_feature_channels


class MuZeroNetwork(_HyperMz):
    @torch.jit.export
    def get_type_name(self) -> str: return "muzero"


class MuZeroAtariNetwork(_HyperMz):
    @torch.jit.export
    def get_type_name(self) -> str: return "muzero_atari"


_KINDS = {"alphazero": (AlphaZeroNetwork, _tree_alphazero), "muzero": (MuZeroNetwork, _tree_muzero), "muzero_atari": (MuZeroAtariNetwork, _tree_muzero_atari)}
_FIELDS = ["num_input_channels", "input_channel_height", "input_channel_width", "num_hidden_channels", "hidden_channel_height", "hidden_channel_width",
           "num_action_feature_channels", "num_blocks", "action_size", "num_value_hidden_channels", "discrete_value_size"]


def build_module(desc):
    """desc: the worker's mz_net_desc (ctypes, minizero_amd.lib.NetDesc)"""
    d = {k: int(getattr(desc, k)) for k in _FIELDS}
    d["game_name"] = desc.game_name.decode()
    kind = {0: "alphazero", 1: "muzero", 2: "muzero_atari"}[int(desc.type)]
    cls, tree = _KINDS[kind]
    return cls(d, tree(d)).eval()


def fill_from_blob(module, blob):
    """blob: every floating tensor of the state_dict in order (the mz_net_create layout, minizero_amd/export_weights.py)"""
    blob = np.ascontiguousarray(blob, np.float32)
    at = 0
    with torch.no_grad():
        for k, t in module.state_dict().items():
            if k.endswith("num_batches_tracked"):
                continue
            n = t.numel()
            t.copy_(torch.from_numpy(blob[at:at + n].reshape(tuple(t.shape)).copy()))
            at += n
    assert at == blob.size, (at, blob.size)


def write_pt(path, desc, blob):
    m = build_module(desc)
    fill_from_blob(m, blob)
    torch.jit.script(m).save(path)
    return path
