#!/bin/bash
# Per-KERNEL counters of one run_configs.py leg (default: c5, BASELINE configs[4]) — the aggregated r0N_pmc_<leg>.json of refresh_profiles.sh sums every
# simulation-side kernel into one row; here every kernel keeps its own: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, one pass each, corrected as
# MI355X_MICROARCH.md prescribes for gfx950), MFMA-busy share, LDS bank-conflict share (third pass).  Counter passes carry --kernel-trace only.
#   usage (GPU box): bash tools/pmc_per_kernel.sh [leg]      -> gpurun_out/pmck/pmc_<leg>_per_kernel.json (+ .txt table)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
K=${1:-c5}
O=gpurun_out/pmck
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/a -- python tools/run_configs.py $K --out $O/tmp.json > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/b -- python tools/run_configs.py $K --out $O/tmp.json > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c -- python tools/run_configs.py $K --out $O/tmp.json > $O/c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python tools/run_configs.py $K --out $O/tmp.json > $O/s.log 2>&1
python - "$K" <<'PY'
import collections, csv, glob, json, re, sys
K = sys.argv[1]
O = "gpurun_out/pmck"
def short(name):
    n = name.split("(")[0].replace("void ", "").replace("mz::", "")
    return re.sub(r"\s+", "", n)
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return acc, n
fa, na = load("a"); fb, _ = load("b"); fc, _ = load("c")
dur = {}
for f in glob.glob(f"{O}/s/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
rows = {}
for k in sorted(set(fa) | set(fc) | set(dur)):
    d = max(1, na[k]["FETCH_SIZE"]) if k in na else 1
    row = {"launches_counted": d}
    if k in fa and k in fb:
        row["hbm_bytes_per_launch"] = (2.0 * fa[k]["FETCH_SIZE"] + fb[k]["WRITE_SIZE"]) * 1024.0 / d
        row["fetch_bytes_per_launch"] = 2.0 * fa[k]["FETCH_SIZE"] * 1024.0 / d
        row["write_bytes_per_launch"] = fb[k]["WRITE_SIZE"] * 1024.0 / d
    if k in fc:
        s = fc[k]
        row["mfma_busy_frac_of_all_simd_cycles"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, s["GRBM_GUI_ACTIVE"] * 128.0)
        row["lds_conflict_frac"] = s["SQ_LDS_BANK_CONFLICT"] / max(1.0, s["SQ_LDS_IDX_ACTIVE"])
        row["lds_active_cycles_per_launch"] = s["SQ_LDS_IDX_ACTIVE"] / d
    if k in dur: row.update({"stats_" + kk: vv for kk, vv in dur[k].items()})
    rows[k] = row
out = {"leg": K, "command": f"python tools/run_configs.py {K}", "note": "per kernel over every launch of the run (warm-up included); HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB "
       "(gfx950: FETCH_SIZE reports half of the bytes, MI355X_MICROARCH.md HBM section); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128), the same normalisation as r0N_pmc_*.json; "
       "lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; stats_* from a separate --kernel-trace --stats run (no counters)", "kernels": rows}
json.dump(out, open(f"{O}/pmc_{K}_per_kernel.json", "w"), indent=1)
with open(f"{O}/pmc_{K}_per_kernel.txt", "w") as f:
    f.write("%-46s %8s %9s %10s %10s %9s %9s\n" % ("kernel", "calls", "avg us", "% of GPU", "HBM KB/l", "MFMA busy", "LDS confl"))
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("stats_total_ms", 0.0)):
        f.write("%-46s %8d %9.1f %10.2f %10.1f %9.3f %9.3f\n" % (k[:46], r.get("stats_calls", 0), r.get("stats_avg_us", 0.0), r.get("stats_pct", 0.0),
                r.get("hbm_bytes_per_launch", 0.0) / 1024.0, r.get("mfma_busy_frac_of_all_simd_cycles", 0.0), r.get("lds_conflict_frac", 0.0)))
print(open(f"{O}/pmc_{K}_per_kernel.txt").read())
PY
rm -rf $O/a $O/b $O/c $O/s
