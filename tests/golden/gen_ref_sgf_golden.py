#!/usr/bin/env python3
"""Outputs of the REFERENCE's own utils/sgf_loader.{h,cpp}, utils/vector_map.h and environment/go/go_unit.h (compiled in place into
oracle/_ref/libmzref.so by `make -C oracle ref`, build container only) -> tests/golden/ref_sgf_vectormap.json: the record state machine
(tags / moves / per-move info / escapes / malformed input), the four coordinate conversions, the insertion-ordered tag map, constants.

usage: make -C oracle ref && python tests/golden/gen_ref_sgf_golden.py
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def sgf_inputs():
    rng = np.random.default_rng(11)
    out = [
        "(;GM[go_9x9]RE[1]OBS[]SZ[9]KM[7.500000]EV[weight_iter_3.pt]DLEN[0-3];B[ee]P[40:200,41:100.5]V[0.12]R[0];W[dd]P[30:1]V[-0.5]R[0];B[]V[0];W[ai])",
        "(;SZ[19]GM[x];B[aa];W[ss];B[js]C[a \\] bracket \\\\ and \\( paren])",
        "(;SZ[9];B[ab]",                                   # no closing paren
        "(;GM[go];B[aa])",                                  # move before SZ: board size unknown
        "garbage (;SZ[5]A[1]A[2]B[x y];W[bb]K[v]K[w];B[cc])",  # repeated keys, spaces in a key position / value
        "(;SZ[9]) trailing",
        "",
        "(SZ[9];B[aa])",
        "(;SZ[13]X[];B[mm]P[];W[am]L[2]P[1:2]L[1])",
    ]
    letters = "abcdefghijklmnopqrs"
    for _ in range(40):
        n = int(rng.choice([5, 9, 13, 19]))
        s = f"(;GM[go_{n}x{n}]SZ[{n}]RE[{rng.integers(-1, 2)}]"
        for m in range(int(rng.integers(0, 12))):
            mv = "" if rng.random() < 0.15 else letters[rng.integers(0, n)] + letters[rng.integers(0, n)]
            s += f";{'BW'[m % 2]}[{mv}]"
            for k in rng.choice(["P", "V", "R", "L", "C"], int(rng.integers(0, 4)), replace=False):
                val = "".join(rng.choice(list("0123456789:,.-]\\ ab("), int(rng.integers(0, 8))))
                val = val.replace("\\", "\\\\").replace("]", "\\]")
                s += f"{k}[{val}]"
        out.append(s + ")")
    return out


def tagmap_ops():
    rng = np.random.default_rng(5)
    out = ["set GM go\nset RE 0\nset RE 1.000000\nset OBS x\nset SZ 9\nset EV w.pt\nset RE -1\nset DLEN 0-5",
           "insert GM a\ninsert RE 0\ninsert RE 7\nset RE 2\nerase GM\ninsert GM b\nset L 1\nerase Q"]
    for _ in range(30):
        ops = []
        for _k in range(int(rng.integers(1, 20))):
            ops.append(f"{rng.choice(['set', 'insert', 'erase'])} {rng.choice(list('ABCDE'))} {int(rng.integers(0, 100))}")
        out.append("\n".join(ops))
    return out


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libmzref.so"))
    buf = C.create_string_buffer(1 << 16)
    fx = {"sgf": [], "tagmap": [], "coords": [], "strings": []}
    for s in sgf_inputs():
        L.mzref_sgf_parse(s.encode(), buf, len(buf))
        fx["sgf"].append({"in": s, "out": buf.value.decode(errors="replace")})
    for ops in tagmap_ops():
        L.mzref_tagmap_apply(ops.encode(), buf, len(buf))
        fx["tagmap"].append({"ops": ops, "out": buf.value.decode()})
    for n in (3, 9, 13, 19):
        for coord, sgf in (("A1", "aa"), ("J9", "ii"), ("H8", "hh"), ("T19", "ss"), ("pass", ""), ("PASS", "tt"), ("c3", "cc"), ("K10", "ka"), ("Z", "a"), ("b12", "abc")):
            o = (C.c_int * 2)()
            L.mzref_sgf_coords(0, n, coord.encode(), sgf.encode(), o)
            fx["coords"].append({"n": n, "coord": coord, "sgf": sgf, "out": [o[0], o[1]]})
        for a in sorted(set([0, 1, n - 1, n, n * n - 1, n * n, (n * n) // 2, 8, 9])):
            if a <= n * n:
                L.mzref_sgf_strings(a, n, buf, len(buf))
                fx["strings"].append({"n": n, "action": a, "out": buf.value.decode()})
    o = (C.c_int * 4)()
    name_len = L.mzref_go_constants(o)
    fx["go_unit"] = {"kMaxGoBoardSize": o[0], "kGoNumPlayer": o[1], "sizeof_GoHashKey": o[2], "GoBitboard_bits": o[3], "kGoName_len": name_len}
    path = os.path.join(HERE, "ref_sgf_vectormap.json")
    json.dump(fx, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", len(fx["sgf"]), "records,", len(fx["tagmap"]), "tag-map sequences")


if __name__ == "__main__":
    main()
