set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/p2; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python tools/run_configs.py c5 --moves 10 --out $O/c5_prof.json > $O/tr.log 2>&1
python tools/c5_move_timeline.py $(find $O/tr -name "*kernel_trace.csv" | head -1) > $O/timeline.txt 2>&1
cp $(find $O/tr -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/tr
timeout 200 python tools/run_configs.py c5 --out $O/c5.json 2>&1 | cut -c1-200 > $O/c5.log
