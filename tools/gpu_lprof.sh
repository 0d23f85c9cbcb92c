#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/lprof; rm -rf $O; mkdir -p $O
touch minizero_amd/csrc/sim.hip
make -s -C minizero_amd/csrc EXTRA="-DMZ_SIM_LPROF" 2>&1 | tail -3
MZ_SIM_PROF=1 timeout 300 python bench.py --steps 4 --warmup 1 --game-moves 0 --no-cpu-baseline 2>&1 | grep "mz sim" | tee $O/lprof.txt
