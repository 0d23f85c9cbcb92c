#!/usr/bin/env python3
"""Generate NN golden vectors from the REFERENCE's own Python modules (minizero/network/py).

Runs ONLY in the build container (needs /root/reference); the reference code never travels.
What is committed is data: for each config the generator seed, the input seed/shape and the f32-CPU
outputs of the reference module (a few KB each, nn_<cfg>.npz).  Weights are NOT stored: they are
regenerated from the repo's deterministic generator (same SplitMix64 stream in oracle/o_nn.cpp,
minizero_amd/csrc/weights.cpp and below) and loaded into the reference module with load_state_dict.

usage: python tests/golden/gen_nn_golden.py [name ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(HERE))
from minizero.network.py.create_network import create_network  # noqa: E402  (the reference)
import oracle_lib as O  # noqa: E402

M64 = (1 << 64) - 1


def mix64(z):
    z = np.asarray(z, np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def counter_u01(seed, n):
    """u in [0,1) with 24 bits, identical to the C generator: mix64(seed + (i+1)*GOLDEN) >> 40) * 2^-24."""
    with np.errstate(over="ignore"):
        idx = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed & M64)
        z = mix64(idx)
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def gen_weights_numpy(module, seed):
    """Python twin of Net::generateRaw (oracle/o_nn.cpp): state_dict order, num_batches_tracked skipped."""
    specs = []
    for name, t in module.state_dict().items():
        if name.endswith("num_batches_tracked"):
            continue
        n = t.numel()
        if name.endswith("running_var") or (name.endswith("weight") and t.dim() == 1):
            lo, hi = np.float32(0.5), np.float32(1.5)
        elif name.endswith("running_mean") or (name.endswith("bias") and ".bn" in "." + name.rsplit(".", 1)[0].split(".")[-1]):
            lo, hi = np.float32(-0.1), np.float32(0.1)
        else:  # conv / linear weight or bias: +-1/sqrt(fan_in)
            w = module.state_dict()[name.rsplit(".", 1)[0] + ".weight"]
            fan_in = int(np.prod(w.shape[1:]))
            bound = np.float32(1.0) / np.sqrt(np.float32(fan_in))
            lo, hi = -bound, bound
        specs.append((name, tuple(t.shape), n, lo, hi))
    total = sum(s[2] for s in specs)
    u = counter_u01(seed, total)
    blob = np.empty(total, np.float32)
    off = 0
    for _, _, n, lo, hi in specs:
        blob[off:off + n] = lo + (hi - lo) * u[off:off + n]
        off += n
    return blob, specs


def load_blob(module, blob, specs):
    sd = module.state_dict()
    off = 0
    for name, shape, n, _, _ in specs:
        sd[name] = torch.from_numpy(blob[off:off + n].reshape(shape).copy())
        off += n
    module.load_state_dict(sd)


def binary_planes(seed, shape):
    n = int(np.prod(shape))
    return (counter_u01(seed, n) < 0.3).astype(np.float32).reshape(shape)


CONFIGS = {
    # name: (create_network args in the reference's order, type)
    "c1_tictactoe_az": ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero"),
    "c2_go_az": ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "alphazero"),
    "c3_othello_az": ("othello_8x8", 4, 8, 8, 64, 8, 8, 1, 6, 65, 256, 1, "alphazero"),
    "c4_go_mz": ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "muzero"),
    "small_go_az": ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero"),
    # BASELINE configs[4] network (the reference picks MuZeroAtariNetwork because "atari" is in the game name)
    "c5_atari_mz": ("atari_ms_pacman", 32, 96, 96, 64, 6, 6, 18, 6, 18, 256, 601, "muzero"),
    "small_atari_mz": ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero"),
    # round 5: shapes beyond BASELINE.json — the reference's defaults (config/configuration.cpp:70-72: 1 block x 256 channels), wider / larger boards
    # (go_unit.h:11: up to 19x19), channel counts that are no multiple of 16 (the run-time-shaped kernels)
    "w_go9_1bx256_az": ("go_9x9", 18, 9, 9, 256, 9, 9, 1, 1, 82, 256, 1, "alphazero"),
    "w_go9_6bx128_az": ("go_9x9", 18, 9, 9, 128, 9, 9, 1, 6, 82, 256, 1, "alphazero"),
    "w_go19_6bx64_az": ("go_19x19", 18, 19, 19, 64, 19, 19, 1, 6, 362, 256, 1, "alphazero"),
    "w_go7_2bx32_az": ("go_7x7", 18, 7, 7, 32, 7, 7, 1, 2, 50, 256, 1, "alphazero"),
    "w_go13_2bx96_az": ("go_13x13", 18, 13, 13, 96, 13, 13, 1, 2, 170, 64, 1, "alphazero"),
    "w_go5_3bx24_az": ("go_5x5", 18, 5, 5, 24, 5, 5, 1, 3, 26, 20, 1, "alphazero"),
    "w_go9_2bx128_mz": ("go_9x9", 18, 9, 9, 128, 9, 9, 1, 2, 82, 256, 1, "muzero"),
    "w_go7_1bx40_mz": ("go_7x7", 18, 7, 7, 40, 7, 7, 1, 1, 50, 32, 1, "muzero"),
    "w_oth8_1bx256_az": ("othello_8x8", 4, 8, 8, 256, 8, 8, 1, 1, 65, 256, 1, "alphazero"),
    "w_ttt_1bx256_az": ("tictactoe", 4, 3, 3, 256, 3, 3, 1, 1, 9, 256, 1, "alphazero"),
}


def float_planes(seed, shape):
    return counter_u01(seed, int(np.prod(shape))).reshape(shape).astype(np.float32)


def main():
    torch.set_num_threads(1)
    only = set(sys.argv[1:])  # optional: the names to (re)generate
    for name, args in CONFIGS.items():
        if only and name not in only:
            continue
        net = create_network(*args).eval()
        wseed = 0
        blob, specs = gen_weights_numpy(net, wseed)
        # the numpy twin must equal the C generator bit for bit
        atari = "atari" in args[0]
        d = O.make_desc(*args[:10], vh=args[10], dv=args[11], type_name="muzero_atari" if atari else args[12])
        cblob = O.gen_weights(d, wseed)
        assert blob.shape == cblob.shape and np.array_equal(blob.view(np.uint32), cblob.view(np.uint32)), name
        load_blob(net, blob, specs)
        out = {"create_network_args": np.array([str(a) for a in args]), "weight_seed": wseed}
        with torch.no_grad():
            for B in ((1, 2) if atari else (1, 3)):
                iseed = 1000 + B
                x = (float_planes if atari else binary_planes)(iseed, (B, args[1], args[2], args[3]))
                if args[12] == "alphazero":
                    r = net(torch.from_numpy(x))
                    out[f"b{B}_input_seed"] = iseed
                    out[f"b{B}_policy"] = r["policy"].numpy()
                    out[f"b{B}_policy_logit"] = r["policy_logit"].numpy()
                    out[f"b{B}_value"] = r["value"].numpy().reshape(-1)
                else:
                    r = net.initial_inference(torch.from_numpy(x))
                    out[f"b{B}_input_seed"] = iseed
                    for k in ("policy", "policy_logit", "hidden_state"):
                        out[f"b{B}_init_{k}"] = r[k].numpy().reshape(B, -1)
                    out[f"b{B}_init_value"] = r["value"].numpy().reshape(-1)
                    # recurrent step on the reference's own hidden state with a one-hot action plane
                    act = np.zeros((B, args[7], args[5], args[6]), np.float32)
                    for b in range(B):
                        if atari:
                            act[b, (7 * b + 3) % args[7]] = 1.0  # plane `action` all ones (ref atari.cpp:124-130)
                        else:
                            act[b, 0].reshape(-1)[(7 * b + 3) % (args[5] * args[6])] = 1.0
                    r2 = net.recurrent_inference(r["hidden_state"], torch.from_numpy(act))
                    for k in ("policy", "policy_logit", "hidden_state"):
                        out[f"b{B}_rec_{k}"] = r2[k].numpy().reshape(B, -1)
                    if atari:  # 601-bin distributions: store the expectation in the transformed space (f64 sum of the reference's f32 probabilities)
                        bins = np.arange(-(args[11] // 2), args[11] // 2 + 1, dtype=np.float64)
                        out[f"b{B}_init_value"] = (r["value"].numpy().astype(np.float64) * bins).sum(1).astype(np.float32)
                        out[f"b{B}_rec_value"] = (r2["value"].numpy().astype(np.float64) * bins).sum(1).astype(np.float32)
                        out[f"b{B}_rec_reward"] = (r2["reward"].numpy().astype(np.float64) * bins).sum(1).astype(np.float32)
                    else:
                        out[f"b{B}_rec_value"] = r2["value"].numpy().reshape(-1)
        path = os.path.join(HERE, f"nn_{name}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes; params", blob.size)


if __name__ == "__main__":
    main()
