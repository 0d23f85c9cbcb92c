"""Weight files for the worker (off the hot path; torch is only used here when converting a trained model).

`.mzw` = b"MZW1" + mz_net_desc (112 bytes, include/mzgpu.h) + uint64 count + count f32 values: every floating tensor
of the module's state_dict() in order (num_batches_tracked skipped) — the layout `mz_net_create` takes.

    python -m minizero_amd.export_weights weight_iter_100.pt          # TorchScript file written by the reference's train.py
    python -m minizero_amd.export_weights --synthetic c2 out.mzw      # deterministic synthetic weights (bench / tests)

One-line hook for the reference's trainer (ref learner/train.py:121-127, next to torch.jit.script(...).save(path)):
    from minizero_amd.export_weights import export_module; export_module(self.network, path[:-3] + ".mzw")
"""
import ctypes as C
import struct
import sys

import numpy as np

from .lib import NetDesc, make_desc, DESCS, generate_weights


def write_mzw(path, desc, weights):
    w = np.ascontiguousarray(weights, np.float32)
    with open(path, "wb") as f:
        f.write(b"MZW1")
        f.write(bytes(desc))
        f.write(struct.pack("<Q", w.size))
        f.write(w.tobytes())


def read_mzw(path):
    with open(path, "rb") as f:
        assert f.read(4) == b"MZW1"
        desc = NetDesc.from_buffer_copy(f.read(C.sizeof(NetDesc)))
        (n,) = struct.unpack("<Q", f.read(8))
        return desc, np.frombuffer(f.read(4 * n), np.float32).copy()


def desc_of_module(m, type_name=None):
    """the 12 hyper-parameters the reference exports as TorchScript methods (ref alphazero_network.py:42-88)"""
    g = lambda name, default=None: (getattr(m, name)() if callable(getattr(m, name, None)) else default)  # noqa: E731
    tn = type_name or g("get_type_name")
    ac = g("get_num_action_feature_channels", 1) if tn != "alphazero" else 1
    return make_desc(g("get_game_name"), g("get_num_input_channels"), g("get_input_channel_height"), g("get_input_channel_width"),
                     g("get_num_hidden_channels"), g("get_hidden_channel_height"), g("get_hidden_channel_width"), ac, g("get_num_blocks"),
                     g("get_action_size"), g("get_num_value_hidden_channels"), g("get_discrete_value_size"), tn)


def blob_of_state_dict(sd):
    parts = [t.detach().cpu().float().reshape(-1).numpy() for k, t in sd.items() if not k.endswith("num_batches_tracked")]
    return np.concatenate(parts).astype(np.float32)


def export_module(module, path):
    write_mzw(path, desc_of_module(module), blob_of_state_dict(module.state_dict()))


def main(argv):
    if len(argv) >= 3 and argv[0] == "--synthetic":
        d = DESCS[argv[1]]()
        write_mzw(argv[2], d, generate_weights(d, int(argv[3]) if len(argv) > 3 else 0))
        return 0
    import torch
    for pt in argv:
        m = torch.jit.load(pt, map_location="cpu")
        export_module(m, pt[:-3] + ".mzw" if pt.endswith(".pt") else pt + ".mzw")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
