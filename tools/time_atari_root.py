#!/usr/bin/env python3
"""The root representation of BASELINE configs[4] (96x96 -> 6x6, 64 games) on its own: `rocprofv3 --kernel-trace --stats -- python tools/time_atari_root.py`
gives the per-kernel averages of conv3x3_tiled / avgpool / tower / heads without the search around them; the script prints the wall time per call
(which includes the 75 MB of features going over PCIe: the kernel trace is the measurement) and a checksum of the outputs (same weights, same
input: two builds of the library must print the same one)."""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
d = mz.DESCS["c5"]()
net = mz.Net(d, mz.generate_weights(d, 0))
rng = np.random.default_rng(1)
feat = rng.random((B, d.num_input_channels, d.input_channel_height, d.input_channel_width), dtype=np.float32)
out = net.initial_inference(feat)
t0 = time.time()
for _ in range(iters):
    out = net.initial_inference(feat)
dt = (time.time() - t0) / iters
crc = 0
for a in out:
    crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
print("initial_inference B=%d: %.2f ms per call (host copies included), outputs crc32 %08x" % (B, dt * 1e3, crc))
