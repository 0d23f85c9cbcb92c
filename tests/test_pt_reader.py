"""The native TorchScript reader (ptfile.cpp: zip directory + the pickle subset torch.jit emits) against torch itself — CPU only.

The files are scripted at test time from the REFERENCE's own Python network modules (exactly what learner/train.py:127 saves), so the
test needs /root/reference and torch; a .pt embeds the reference's module source, so none is committed as a fixture."""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "minizero", "network", "py")), reason="needs the reference's Python network modules")

CASES = {
    "small_go_az": ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero"),
    "c1_tictactoe_az": ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"),
    "c2_go_az": ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "alphazero"),
    "small_go_mz": ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero"),
    "small_atari_mz": ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_native_reader_matches_torch(mz, name, tmp_path):
    import torch
    sys.path.insert(0, REF)
    from minizero.network.py.create_network import create_network
    from minizero_amd.export_weights import blob_of_state_dict, desc_of_module
    torch.manual_seed(hash(name) % 1000)
    net = create_network(*CASES[name])
    with torch.no_grad():  # non-trivial running statistics
        for k, t in net.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                t.copy_(torch.rand_like(t) + 0.5)
    net.eval()
    path = str(tmp_path / "weight_iter_100.pt")
    torch.jit.script(net).save(path)
    desc, w = mz.read_pt(path)
    want = blob_of_state_dict(net.state_dict())
    assert w.shape == want.shape and np.array_equal(w.view(np.uint32), want.view(np.uint32))
    ref = desc_of_module(torch.jit.load(path, map_location="cpu"))
    assert bytes(desc) == bytes(ref), (desc.game_name, desc.type, ref.type)
    assert mz.param_count(desc) == len(w)


def test_reader_rejects_garbage(mz, tmp_path):
    p = tmp_path / "x.pt"
    p.write_bytes(b"not a zip archive at all" * 10)
    with pytest.raises(mz.MzError):
        mz.read_pt(str(p))
    with pytest.raises(mz.MzError):
        mz.read_pt(str(tmp_path / "missing.pt"))


def test_reader_survives_corrupted_archives(mz, tmp_path):
    """Crafted / damaged files must come back as MzError, never as an out-of-bounds read: truncations, flipped bytes in the zip directory and
    in the pickle (sizes / strides / offsets), 0xFF-filled ZIP64-style fields."""
    import torch
    sys.path.insert(0, REF)
    from minizero.network.py.create_network import create_network
    net = create_network(*CASES["small_go_az"])
    net.eval()
    path = str(tmp_path / "good.pt")
    torch.jit.script(net).save(path)
    good = open(path, "rb").read()
    desc, w = mz.read_pt(path)
    rng = np.random.default_rng(0)
    bad = tmp_path / "bad.pt"
    pkl = good.find(b"data.pkl")
    outcomes = {"ok": 0, "rejected": 0}
    variants = [good[:n] for n in (10, 100, len(good) // 2, len(good) - 30, len(good) - 5)]
    for _ in range(300):
        b = bytearray(good)
        region = rng.integers(0, 3)
        lo, hi = [(max(0, len(b) - 400), len(b)), (pkl, pkl + 4000), (0, len(b))][region]
        for _k in range(int(rng.integers(1, 6))):
            b[int(rng.integers(lo, min(hi, len(b))))] = int(rng.choice([0, 0xFF, rng.integers(0, 256)]))
        variants.append(bytes(b))
    for v in variants:
        bad.write_bytes(v)
        try:
            d2, w2 = mz.read_pt(str(bad))
            outcomes["ok"] += 1
            assert len(w2) == mz.param_count(d2)
        except mz.MzError:
            outcomes["rejected"] += 1
    assert outcomes["rejected"] > 20
