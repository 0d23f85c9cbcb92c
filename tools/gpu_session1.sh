#!/bin/bash
# round-2 session 1: new baseline-net parity tests, the new bench line, rocprof stats of C3/C4/C5
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s1
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_baseline_nets.py -x -q > $O/pytest_baseline.log 2>&1; echo "pytest rc $?" >> $O/pytest_baseline.log
tail -15 $O/pytest_baseline.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 3000 $O/bench.json; tail -5 $O/bench.err
timeout 600 python tools/run_configs.py --out $O/configs.json > $O/configs.log 2>&1; cat $O/configs.log | cut -c1-700
for k in c3 c4 c5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$k -- python tools/run_configs.py $k --out $O/configs_prof_$k.json > $O/stats_$k.log 2>&1
  cp $(find $O/stats_$k -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$k.csv 2>/dev/null
  rm -rf $O/stats_$k
  python tools/kstats.py $O/kernel_stats_$k.csv | head -8
done
