set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/refresh5; rm -rf $O; mkdir -p $O
k=c5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$k -- python tools/run_configs.py $k --out $O/configs_prof_$k.json > $O/stats_$k.log 2>&1
cp $(find $O/stats_$k -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$k.csv
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcA_$k -- python tools/run_configs.py $k --out $O/tmp.json > $O/pmcA_$k.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcB_$k -- python tools/run_configs.py $k --out $O/tmp.json > $O/pmcB_$k.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmcC_$k -- python tools/run_configs.py $k --out $O/tmp.json > $O/pmcC_$k.log 2>&1
python - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/refresh5"
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = "sim_kernel" if "sim_kernel" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0][-40:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("FETCH_SIZE", "GRBM_GUI_ACTIVE"): n[k] += 1
    return acc, n
a, n = load("pmcA_c5"); aw, _ = load("pmcB_c5"); b, _ = load("pmcC_c5")
cycles = (40 + 14) * 51
f, w = a["sim_kernel"]["FETCH_SIZE"], aw["sim_kernel"]["WRITE_SIZE"]; s = b["sim_kernel"]
out = {"kernel": "sim_kernel_mz_cluster<6,6,84,64>", "dispatches": n["sim_kernel"], "lockstep_cycles": cycles, "FETCH_SIZE_KB_total": f, "WRITE_SIZE_KB_total": w,
       "bytes_per_cycle": (2.0 * f + w) * 1024.0 / cycles,
       "note": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE reports half of the bytes, MI355X_MICROARCH.md HBM section; re-calibrated on a copy kernel), over all simulation-kernel launches of `python tools/run_configs.py c5` (%d lock-step cycles of 64 games incl. warm-up)" % cycles,
       "SQ_VALU_MFMA_BUSY_CYCLES": s["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": s["GRBM_GUI_ACTIVE"],
       "mfma_busy_frac_of_all_simd_cycles": s["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, s["GRBM_GUI_ACTIVE"] * 128.0),
       "lds_conflict_frac": s["SQ_LDS_BANK_CONFLICT"] / max(1.0, s["SQ_LDS_IDX_ACTIVE"])}
json.dump(out, open(f"{O}/pmc_c5.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
timeout 200 python tools/run_configs.py c5 --out $O/configs_c5.json 2>&1 | cut -c1-120
rm -rf $O/stats_* $O/pmcA_* $O/pmcB_* $O/pmcC_* $O/tmp.json
