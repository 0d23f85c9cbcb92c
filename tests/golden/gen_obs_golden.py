#!/usr/bin/env python3
"""Golden vectors for utils::compressString (ref utils/utils.h:35-91) -> tests/golden/compress_string.json.

The reference gzips with boost::iostreams::gzip_compressor (default gzip_params) and prints lower-case hex.  Boost is not in this image; its
gzip filter wraps zlib, which is (Python's zlib module = the same libz 1.2.11).  The vectors below are built from the member layout that
filter writes — header 1f 8b 08 00 | mtime 0 | xfl 0 | os ff, raw deflate at level 6 / window 15 / mem level 8, CRC-32 and length little-endian —
spelled out explicitly rather than through gzip.compress (whose header bytes changed between Python versions).  Inputs are described by a
generator name so that the fixture stays small; big outputs are stored as length + SHA-256.
"""
import hashlib
import json
import os
import struct
import zlib

import numpy as np


def compress_string(data: bytes) -> str:
    if not data:
        return ""
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY)
    body = co.compress(data) + co.flush()
    member = b"\x1f\x8b\x08\x00" + struct.pack("<I", 0) + b"\x00\xff" + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)
    return member.hex()


def make_input(kind, n, seed=0):
    if kind == "empty":
        return b""
    if kind == "text":
        return (b"(;GM[atari_ms_pacman]RE[0]SD[12345]" * (n // 35 + 1))[:n]
    if kind == "zeros":
        return bytes(n)
    if kind == "random":
        return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == "blocks":  # frame-like: 8x8 blocks of constant bytes, 96 wide
        rng = np.random.default_rng(seed)
        rows = []
        total = 0
        while total < n:
            vals = rng.integers(0, 256, 12, dtype=np.uint8)
            row = np.repeat(vals, 8).tobytes()
            rows.extend([row] * 8)
            total += 8 * len(row)
        return b"".join(rows)[:n]
    raise ValueError(kind)


CASES = [("empty", 0, 0), ("text", 1, 0), ("text", 35, 0), ("text", 4000, 0), ("zeros", 27648, 0), ("random", 1000, 1), ("random", 70000, 2),
         ("blocks", 27648, 3), ("blocks", 27648 * 20, 4), ("blocks", 27648 * 60 + 17, 5)]

if __name__ == "__main__":
    import gzip
    out = []
    for kind, n, seed in CASES:
        data = make_input(kind, n, seed)
        hx = compress_string(data)
        assert (gzip.decompress(bytes.fromhex(hx)) if hx else b"") == data
        e = {"kind": kind, "n": n, "seed": seed, "hex_len": len(hx), "hex_sha256": hashlib.sha256(hx.encode()).hexdigest()}
        if len(hx) <= 400:
            e["hex"] = hx
        out.append(e)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compress_string.json")
    json.dump({"zlib": zlib.ZLIB_RUNTIME_VERSION, "cases": out}, open(path, "w"), indent=1)
    print("wrote", path, len(out), "cases")
