#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c5prof; rm -rf $O; mkdir -p $O
MZ_SIM_PROF=1 timeout 200 python tools/run_configs.py c5 --out $O/c5.json 2>&1 | grep -v "^c5" | tee $O/prof_plain.txt
touch minizero_amd/csrc/sim.hip
make -s -C minizero_amd/csrc EXTRA="-DMZ_SIM_HPROF" 2>&1 | tail -3
MZ_SIM_PROF=1 timeout 200 python tools/run_configs.py c5 --out $O/c5h.json 2>&1 | grep -v "^c5" | tee $O/prof_hprof.txt
