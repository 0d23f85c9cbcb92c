#!/usr/bin/env python3
"""Dump outputs of the REFERENCE's own random.{h,cpp}, rotation.h and config/*.cpp (compiled in place into
oracle/_ref/libmzref.so by `make -C oracle ref`, build container only) into a small data fixture, so the
oracle can be checked against the real reference on any box:  tests/golden/ref_rng_rotation_config.json

usage: make -C oracle ref && python tests/golden/gen_ref_golden.py
"""
import ctypes as C
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CONF_STRINGS = [
    "",
    "actor_num_simulation=400:zero_num_parallel_games=256:env_board_size=9",
    "actor_use_gumbel=true:actor_use_gumbel_noise=TRUE:actor_use_dirichlet_noise=0:actor_gumbel_sigma_scale_c=0.1:"
    "actor_mcts_value_rescale=1:actor_mcts_reward_discount=0.997:nn_file_name=/a/b/c.pt:program_seed=42",
    "actor_select_action_softmax_temperature = 0.5 # comment:zero_actor_ignored_command=reset_actors keep_alive",
    "no_such_key=1",
    "actor_num_simulation=abc",
    "actor_use_gumbel=maybe",
]


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmzref.so"))
    L.mzref_rng_vector.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_double)]
    L.mzref_config_dump.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    out = {"rng": [], "rotation": {}, "reversed_rotation": [L.mzref_reversed_rotation(r) for r in range(8)], "config": []}
    for seed in (1, 7):
        for kind, k, alpha in ((0, 0, 0.0), (1, 0, 0.0), (2, 9, 0.03), (2, 82, 0.03), (2, 5, 0.3), (3, 16, 0.0), (3, 65, 0.0)):
            n = 64 if kind < 2 else 4 * k
            buf = (C.c_double * n)()
            L.mzref_rng_vector(seed, kind, n, k, alpha, buf)
            out["rng"].append({"seed": seed, "kind": kind, "k": k, "alpha": alpha, "values": [float.hex(v) for v in buf]})
    for n in (3, 8, 9, 19):
        out["rotation"][str(n)] = [[L.mzref_rotate(r, p, n) for p in range(n * n + 1)] for r in range(8)]
    # config::* are process-wide globals in the reference: one fresh process per configuration string
    import subprocess
    import sys
    child = ("import ctypes as C,sys,json\n"
             "L=C.CDLL(sys.argv[1]); L.mzref_config_dump.argtypes=[C.c_char_p,C.c_char_p,C.c_int]\n"
             "b=C.create_string_buffer(1<<16); rc=L.mzref_config_dump(sys.argv[2].encode(),b,len(b))\n"
             "print(json.dumps({'rc_negative': rc<0, 'dump': b.value.decode() if rc>=0 else None}))\n")
    for conf in CONF_STRINGS:
        r = subprocess.run([sys.executable, "-c", child, os.path.join(ROOT, "oracle", "_ref", "libmzref.so"), conf], capture_output=True, text=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        d["conf"] = conf
        out["config"].append(d)
    path = os.path.join(HERE, "ref_rng_rotation_config.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
