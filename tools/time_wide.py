#!/usr/bin/env python3
"""Stand-alone tower timing of the shapes beyond BASELINE.json (net_wide.hip tower_wide / conv3x3_any): us per forward of B samples, TFLOP/s of the 3x3 convolutions,
fraction of the f32-MFMA peak.  usage: time_wide.py [B] [name ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import minizero_amd as mz  # noqa: E402
from helpers import WIDE_NN_CFG  # noqa: E402

PEAK = 157.3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
names = sys.argv[2:] or sorted(WIDE_NN_CFG)
extra = {"c2_go_az": ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "alphazero"),
         "x_go19_2bx128_az": ("go_19x19", 18, 19, 19, 128, 19, 19, 1, 2, 362, 256, 1, "alphazero"),
         # shapes without a fused instance: per-layer run-time-shaped kernels (net_wide.hip conv3x3_band; MZ_NO_CONV_BAND=1: conv3x3_any)
         "x_go19_2bx256_az": ("go_19x19", 18, 19, 19, 256, 19, 19, 1, 2, 362, 256, 1, "alphazero"),
         "x_go13_2bx96_az": ("go_13x13", 18, 13, 13, 96, 13, 13, 1, 2, 170, 256, 1, "alphazero"),
         "x_go11_2bx48_az": ("go_11x11", 18, 11, 11, 48, 11, 11, 1, 2, 122, 64, 1, "alphazero"),
         "x_go9_2bx192_az": ("go_9x9", 18, 9, 9, 192, 9, 9, 1, 2, 82, 256, 1, "alphazero")}
out = {}
for name in names:
    args = WIDE_NN_CFG.get(name) or extra[name]
    d = mz.make_desc(*args[:10], vh=args[10], dv=args[11], type_name=args[12])
    net = mz.Net(d, mz.generate_weights(d, 0))
    ms_fwd, ms_tower, fl = net.time_forward(B, 20)
    out[name] = {"B": B, "us_forward": ms_fwd * 1e3, "us_tower": ms_tower * 1e3, "tflops_tower": fl / (ms_tower * 1e-3) / 1e12, "frac_f32_mfma_peak": fl / (ms_tower * 1e-3) / 1e12 / PEAK}
    print(name, json.dumps(out[name]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "time_wide.json"), "w"), indent=1)
