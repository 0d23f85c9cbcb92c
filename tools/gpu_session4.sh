#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loader.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -40 $O/pytest.log
