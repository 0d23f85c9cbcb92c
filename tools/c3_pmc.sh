cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/c3pmc; rm -rf $O; mkdir -p $O
python tools/run_configs.py c3 c1 --out $O/configs.json 2>&1 | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/a -- python tools/run_configs.py c3 --out $O/tmp.json > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/b -- python tools/run_configs.py c3 --out $O/tmp.json > $O/b.log 2>&1
python - <<'PY'
import csv, glob
O="gpurun_out/c3pmc"
def tot(d, name):
    s=0.0
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sim_kernel" in r["Kernel_Name"] and r["Counter_Name"]==name: s+=float(r["Counter_Value"])
    return s
f,w=tot("a","FETCH_SIZE"),tot("b","WRITE_SIZE")
cyc=(40+3)*17
print("C3 HBM bytes per lock-step cycle: %.2f MB (fetch x2 %.2f MB, write %.2f MB)" % ((2*f+w)*1024/cyc/1e6, 2*f*1024/cyc/1e6, w*1024/cyc/1e6))
PY
rm -rf $O/a $O/b
