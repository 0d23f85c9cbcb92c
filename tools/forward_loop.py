#!/usr/bin/env python3
"""Kernel-only workload for rocprofv3 counter passes: forwards of the C2 network at B=256 (the fused tower kernel) and, for
calibrating FETCH_SIZE / WRITE_SIZE on a known byte count with the same 4-B-per-lane coalesced access pattern, the
dynamics-input copy kernel of the C4 network at B=4096 (reads 4096*64*81*4 B, writes 4096*65*81*4 B per launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz  # noqa: E402

d2 = mz.DESCS["c2"]()
net2 = mz.Net(d2, mz.generate_weights(d2, 0))
print("c2 forward (ms total, ms tower, flops):", net2.time_forward(256, 20))
d4 = mz.DESCS["c4"]()
net4 = mz.Net(d4, mz.generate_weights(d4, 0))
print("c4 recurrent B=4096:", net4.time_forward(4096, 3))
