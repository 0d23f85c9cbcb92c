cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp minizero_amd/libmzgpu.so ab/keep.so
export TMPDIR=/tmp
for v in old skip1 skip2 skip4; do
  cp ab/$v.so minizero_amd/libmzgpu.so
  rm -rf gpurun_out/atari_$v; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/atari_$v -o t -- python tools/time_atari_root.py 64 12 > /dev/null 2>&1
  f=$(find gpurun_out/atari_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if "conv3x3" in n: print("%-60s calls %4s avg %8.1f us" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
cp ab/keep.so minizero_amd/libmzgpu.so
