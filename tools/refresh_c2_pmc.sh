#!/bin/bash
# (round 5) ONLY the headline leg's rocprofv3 passes of refresh_profiles.sh (kernel stats + PMC of `bench.py $SHORT`) -> gpurun_out/refresh/{kernel_stats.csv,pmc_sim.json,sim_prof.txt}
# Regenerates the round's evidence under gpurun_out/refresh/ on the GPU box (copy what should be judged into profiles/):
#   bench.json                default bench line (un-profiled): 20 timed moves + the 164-move games/s leg + cpu_baseline
#   bench_bf16x3.json         the opt-in bf16x3 line (own roofline against the bf16 peak)
#   kernel_stats.csv          rocprofv3 --kernel-trace --stats of `bench.py --steps 4 --warmup 1 --game-moves 0 --no-cpu-baseline` (5 moves = 2005 cycles)
#   pmc_sim.json              FETCH_SIZE / WRITE_SIZE passes of the same command -> HBM bytes per lock-step cycle of sim_kernel
#   pmc_util.json             MFMA-busy / LDS-conflict pass
#   configs.json              all five BASELINE configs with a roofline block each (tools/run_configs.py)
#   kernel_stats_c{3,4,5}.csv + pmc_c{3,4,5}.json   rocprofv3 stats / PMC of the other configs' simulation kernels
#   sim_prof.txt              in-kernel phase profile (MZ_SIM_PROF=1)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/refresh
mkdir -p $O
SHORT="--steps 4 --warmup 1 --game-moves 0 --no-cpu-baseline --other-moves 0 --one-stream-moves 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py $SHORT > $O/bench_profiled.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
# FETCH_SIZE takes 3 of the 4 TCC slots and WRITE_SIZE 2: one pass each (MI355X_MICROARCH.md, PMC section)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc1a -- python bench.py $SHORT > $O/bench_pmc1a.json 2> $O/pmc1a.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc1b -- python bench.py $SHORT > $O/bench_pmc1b.json 2> $O/pmc1b.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc2 -- python bench.py $SHORT > $O/bench_pmc2.json 2> $O/pmc2.err
python - <<'PY'
import csv, glob, json, collections, subprocess, sys
O = "gpurun_out/refresh"
names = {}
FP = subprocess.check_output([sys.executable, "bench.py", "--kernel-fingerprint"], text=True).strip()  # the kernel sources these counters were taken on (bench.py compares)
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    global names
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            # (the simulation kernels of a config: sim_kernel / sim_kernel_mz / sim_kernel_mz_cluster and, with Gumbel rounds, the kernels that evaluate a round's leaves
            # ahead: sim_pre_kernel_mz, sim_pre_pair_kernel_mz, the batched pipeline pre_walk / pre_tower / pre_fc / pre_tail of sim_rounds.hip)
            kn = r["Kernel_Name"]
            k = "sim_kernel" if any(t in kn for t in ("sim_kernel", "sim_pre_kernel", "sim_pre_pair_kernel", "pre_walk_kernel", "pre_tower_kernel", "pre_fc_kernel", "pre_tail_kernel")) else kn.split("(")[0][-40:]
            names.setdefault(k, set()).add(kn.split("(")[0].replace("void mz::", ""))
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("FETCH_SIZE", "GRBM_GUI_ACTIVE"): n[k] += 1
    return acc, n
def summarize(tag, a_dir, b_dir, c_dir, cycles, note):
    a, n = load(a_dir); aw, _ = load(b_dir); b, _ = load(c_dir)
    out = {}
    if "sim_kernel" in a and "sim_kernel" in aw:
        f, w = a["sim_kernel"]["FETCH_SIZE"], aw["sim_kernel"]["WRITE_SIZE"]
        out.update({"kernel": "sim_kernel", "kernel_names": sorted(names.get("sim_kernel", [])), "kernel_fingerprint": FP, "dispatches": n["sim_kernel"], "lockstep_cycles": cycles, "FETCH_SIZE_KB_total": f, "WRITE_SIZE_KB_total": w,
                    "bytes_per_cycle": (2.0 * f + w) * 1024.0 / cycles,
                    "note": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE reports half of the bytes, MI355X_MICROARCH.md HBM section; re-calibrated on a copy kernel), " + note})
    if "sim_kernel" in b:
        s = b["sim_kernel"]
        out.update({"SQ_VALU_MFMA_BUSY_CYCLES": s["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": s["GRBM_GUI_ACTIVE"],
                    "mfma_busy_frac_of_all_simd_cycles": s["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, s["GRBM_GUI_ACTIVE"] * 128.0),
                    "lds_conflict_frac": s["SQ_LDS_BANK_CONFLICT"] / max(1.0, s["SQ_LDS_IDX_ACTIVE"])})
    json.dump(out, open(f"{O}/{tag}.json", "w"), indent=1)
summarize("pmc_sim", "pmc1a", "pmc1b", "pmc2", 5 * 401, "over all launches of `python bench.py --steps 4 --warmup 1 --game-moves 0 --no-cpu-baseline` (5 moves x 401 cycles x 256 games)")
PY
MZ_SIM_PROF=1 timeout 200 python bench.py $SHORT 2>&1 | grep "sim prof" > $O/sim_prof.txt
rm -rf $O/stats $O/pmc1a $O/pmc1b $O/pmc2
cat $O/pmc_sim.json; python tools/kstats.py $O/kernel_stats.csv | head -4
