#!/bin/bash
# conv3x3_band (LDS-staged, run-time-shaped) against conv3x3_any (B operand from global memory) on shapes without a fused instance: parity tests, then the stand-alone
# tower timing with and without MZ_NO_CONV_BAND=1 -> gpurun_out/conv_band/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/conv_band; rm -rf $O; mkdir -p $O
B=${1:-64}
SHAPES="x_go19_2bx128_az x_go19_2bx256_az x_go13_2bx96_az x_go11_2bx48_az x_go9_2bx192_az"
timeout 600 python -m pytest tests/test_gpu_net_wide.py -q -x -m gpu 2>&1 | tail -5 > $O/tests.log; tail -2 $O/tests.log
timeout 300 python tools/time_wide.py $B $SHAPES > $O/band.log 2>&1; cp gpurun_out/time_wide.json $O/band.json
MZ_NO_CONV_BAND=1 timeout 300 python tools/time_wide.py $B $SHAPES > $O/any.log 2>&1; cp gpurun_out/time_wide.json $O/any.json
python - <<'PY'
import json
a=json.load(open("gpurun_out/conv_band/any.json")); b=json.load(open("gpurun_out/conv_band/band.json"))
for k in b: print("%-20s conv3x3_any %8.1f us %6.1f TF/s | conv3x3_band %8.1f us %6.1f TF/s (%.3f of peak)" % (k, a[k]["us_tower"], a[k]["tflops_tower"], b[k]["us_tower"], b[k]["tflops_tower"], b[k]["frac_f32_mfma_peak"]))
PY
