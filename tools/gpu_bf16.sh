#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/bf16; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
