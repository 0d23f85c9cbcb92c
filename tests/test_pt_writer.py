"""tests/pt_writer.py (the repo's own TorchScript writer, used by the -m gpu iteration tests for `load_model <file>.pt`) — CPU only.

With /root/reference present: the module it builds has the state_dict keys, shapes and getter values of the reference's own
`create_network(...)` module for every network type, i.e. the file it saves is laid out like a trainer-written one.  Everywhere: the
native reader (ptfile.cpp) gives back exactly the hyper-parameters and the blob the file was written from."""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
CASES = {
    "ttt_az": ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"),
    "go_az": ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero"),
    "go_az_6bx64": ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "alphazero"),
    "othello_az": ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero"),
    "go_mz": ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero"),
    "atari_mz": ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero"),
    "atari_mz_2b": ("atari_ms_pacman", 32, 96, 96, 64, 6, 6, 18, 2, 18, 24, 601, "muzero"),
}


def _desc(mz, a):
    return mz.make_desc(*a[:10], vh=a[10], dv=a[11], type_name="muzero_atari" if a[0].startswith("atari") else a[12])


@pytest.mark.parametrize("name", sorted(CASES))
def test_written_file_reads_back(mz, name, tmp_path):
    import pt_writer
    d = _desc(mz, CASES[name])
    w = mz.generate_weights(d, 7)
    path = pt_writer.write_pt(str(tmp_path / "weight_iter_7.pt"), d, w)
    d2, w2 = mz.read_pt(path)
    assert bytes(d2) == bytes(d) and np.array_equal(w2.view(np.uint32), w.view(np.uint32))
    d3, w3 = mz.read_weights_once(path)
    assert bytes(d3) == bytes(d) and np.array_equal(w3.view(np.uint32), w.view(np.uint32))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "minizero", "network", "py")), reason="needs the reference's Python network modules")
@pytest.mark.parametrize("name", sorted(CASES))
def test_own_writer_has_the_reference_layout(mz, name, tmp_path):
    import torch
    import pt_writer
    sys.path.insert(0, REF)
    from minizero.network.py.create_network import create_network
    a = CASES[name]
    ref = create_network(*a)
    own = pt_writer.build_module(_desc(mz, a))
    assert [(k, tuple(t.shape), t.dtype) for k, t in own.state_dict().items()] == [(k, tuple(t.shape), t.dtype) for k, t in ref.state_dict().items()]
    own_s, ref_s = torch.jit.script(own), torch.jit.script(ref)
    getters = ["get_type_name", "get_game_name", "get_num_input_channels", "get_input_channel_height", "get_input_channel_width", "get_num_hidden_channels",
               "get_hidden_channel_height", "get_hidden_channel_width", "get_num_blocks", "get_action_size", "get_num_value_hidden_channels",
               "get_discrete_value_size"] + ([] if a[12] == "alphazero" else ["get_num_action_feature_channels"])
    for g in getters:
        assert getattr(own_s, g)() == getattr(ref_s, g)(), g
    assert type(own).__name__ == type(ref).__name__
