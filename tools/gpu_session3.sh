#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "atari or c5 or facade or protocol" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python tools/run_configs.py c5 --moves 210 --out $O/c5.json 2>&1 | cut -c1-400
