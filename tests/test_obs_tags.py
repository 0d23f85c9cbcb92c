"""CPU tests of the Atari record tags: `OBS[...]` = utils::compressString of the observation bytes (gzip member as lower-case
hex, ref utils/utils.h:35-91, base_env.h:216-220) and `L[lives]` (ref atari.cpp:187-197).  The oracle's and the product's compressString are
checked against the committed vectors (tests/golden/compress_string.json, made by gen_obs_golden.py) and through Python's gzip module."""
import ctypes as C
import gzip
import hashlib
import json
import os
import re

import numpy as np

from gen_obs_golden import make_input

HERE = os.path.dirname(os.path.abspath(__file__))


def _product_compress(mz, data):
    L = mz.load()
    L.mz_compress_string.restype = C.c_long
    L.mz_compress_string.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    n = L.mz_compress_string(data, len(data), None, 0)
    assert n >= 0
    buf = C.create_string_buffer(n + 1)
    assert L.mz_compress_string(data, len(data), buf, n + 1) == n
    return buf.value.decode()


def test_compress_string_matches_the_golden_vectors(mz, oracle):
    gold = json.load(open(os.path.join(HERE, "golden", "compress_string.json")))
    assert len(gold["cases"]) >= 8
    for e in gold["cases"]:
        data = make_input(e["kind"], e["n"], e["seed"])
        for name, hx in (("oracle", oracle.compress_string(data)), ("product", _product_compress(mz, data))):
            assert len(hx) == e["hex_len"] and hashlib.sha256(hx.encode()).hexdigest() == e["hex_sha256"], (name, e["kind"], e["n"])
            if "hex" in e:
                assert hx == e["hex"]
            assert hx == hx.lower() and (gzip.decompress(bytes.fromhex(hx)) if hx else b"") == data  # decompressString round trip (utils.h:66-91)
            if hx:
                assert hx.startswith("1f8b0800000000" + "0000ff")


def test_oracle_atari_records_carry_obs_and_lives(oracle):
    """The oracle's Atari-shaped games: OBS decodes to (kept observations) x 3 x 96 x 96 bytes of 8x8-block screens; L[k] tags appear where the
    synthetic lives counter drops, also on moves whose P/V/R were already cleared."""
    conf = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=2:actor_use_dirichlet_noise=false:zero_actor_intermediate_sequence_length=4:"
            "learner_n_step_return=1:learner_muzero_unrolling_step=1:env_atari_episode_length=40:zero_num_parallel_games=2:program_seed=3:nn_file_name=a.pt")
    d = oracle.make_desc("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")
    g = oracle.OracleGroup(conf + ":zero_num_threads=1", d, oracle.gen_weights(d, 0))
    g.cycles(3 * 45)
    lines = g.lines()
    assert any(l.startswith("SelfPlay false") for l in lines) and any(l.startswith("SelfPlay true") for l in lines)
    frame = 3 * 96 * 96
    saw_l = 0
    for l in lines:
        obs = re.search(r"OBS\[([0-9a-f]*)\]", l).group(1)
        raw = gzip.decompress(bytes.fromhex(obs))
        game_len = int(l.split(" ")[3])
        assert len(raw) % frame == 0 and 1 <= len(raw) // frame <= min(game_len + 1, 4 + 8 + 1 + 1 + 1)
        screens = np.frombuffer(raw, np.uint8).reshape(-1, 3, 12, 8, 12, 8)
        assert (screens == screens[:, :, :, :1, :, :1]).all(), "synthetic screens are constant on 8x8 blocks"
        saw_l += len(re.findall(r"L\[[0-3]\]", l))
        assert " " not in l.split(" ", 5)[5][:-2]
    assert saw_l > 0
