cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp minizero_amd/libmzgpu.so ab/keep.so
export TMPDIR=/tmp
for v in old new; do
  cp ab/$v.so minizero_amd/libmzgpu.so
  python tools/time_atari_root.py 64 12
  rm -rf gpurun_out/atari_$v; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/atari_$v -o t -- python tools/time_atari_root.py 64 12 > /dev/null 2>&1
  f=$(find gpurun_out/atari_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    n=r["Name"]; 
    if any(k in n for k in ("conv3x3","avgpool","tower_fused","heads_atari")):
        print("%-60s calls %4s avg %8.1f us" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3)); tot+=float(r["AverageNs"])/1e3*(int(r["Calls"])/13.0)
print("sum per call: %.1f us" % tot)
PY
done
cp ab/keep.so minizero_amd/libmzgpu.so
