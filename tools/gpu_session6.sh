#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s6; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
MZ_SIM_PROF=1 timeout 300 python bench.py --steps 4 --warmup 1 --game-moves 0 --no-cpu-baseline 2>&1 | grep "mz sim prof" | tee $O/sim_prof_f32.txt
