#!/bin/bash
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
rm -rf $O; mkdir -p $O
SHORT="--steps 4 --warmup 1 --game-moves 0 --no-cpu-baseline --other-moves 0 --one-stream-moves 0"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --precision bf16x3 --no-cpu-baseline --other-moves 0 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py $SHORT > $O/bench_profiled.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
# FETCH_SIZE takes 3 of the 4 TCC slots and WRITE_SIZE 2: one pass each (MI355X_MICROARCH.md, PMC section)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc1a -- python bench.py $SHORT > $O/bench_pmc1a.json 2> $O/pmc1a.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc1b -- python bench.py $SHORT > $O/bench_pmc1b.json 2> $O/pmc1b.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc2 -- python bench.py $SHORT > $O/bench_pmc2.json 2> $O/pmc2.err
for k in c3 c4 c5 w9x128 w19x64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$k -- python tools/run_configs.py $k --out $O/configs_prof_$k.json > $O/stats_$k.log 2>&1
  cp $(find $O/stats_$k -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$k.csv 2>/dev/null
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcA_$k -- python tools/run_configs.py $k --out $O/tmp.json > $O/pmcA_$k.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcB_$k -- python tools/run_configs.py $k --out $O/tmp.json > $O/pmcB_$k.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmcC_$k -- python tools/run_configs.py $k --out $O/tmp.json > $O/pmcC_$k.log 2>&1
done
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
MOVES = {"c3": (40 + 3) * 17, "c4": (20 + 3) * 51, "c5": (40 + 14) * 51, "w9x128": (3 + 1) * 401, "w19x64": (2 + 1) * 401}
for k, cyc in MOVES.items():
    summarize(f"pmc_{k}", f"pmcA_{k}", f"pmcB_{k}", f"pmcC_{k}", cyc, f"over all simulation-kernel launches of `python tools/run_configs.py {k}` ({cyc} lock-step cycles incl. warm-up)")
PY
timeout 900 python tools/run_configs.py --out $O/configs.json > $O/configs.log 2>&1
# round 5: shapes beyond BASELINE.json on the one-tile tower (sim_kernel_wide): a roofline block each, the stand-alone towers, the in-kernel phase profile
timeout 600 python tools/run_configs.py w9x128 w9x256 w19x64 w9x128mz --out $O/wide_configs.json > $O/wide_configs.log 2>&1
timeout 300 python tools/run_configs.py c5x512 --out $O/c5_512_games_one_gpu.json > /dev/null 2>&1
timeout 300 python tools/time_wide.py 256 > $O/time_wide.log 2>&1; cp gpurun_out/time_wide.json $O/time_wide_towers.json
for k in w9x128 w19x64; do MZ_SIM_PROF=1 timeout 200 python tools/run_configs.py $k --moves 1 --out $O/tmp.json 2>&1 | grep "sim prof" > $O/sim_prof_$k.txt; done
MZ_SIM_PROF=1 timeout 200 python bench.py $SHORT 2>&1 | grep "sim prof" > $O/sim_prof.txt
MZ_SIM_PROF=1 timeout 200 python tools/run_configs.py c5 --out $O/tmp.json 2>&1 | grep "sim prof" > $O/sim_prof_c5_rounds.txt
MZ_SIM_PROF=1 timeout 200 python tools/run_configs.py c5 --conf mz_sim_rounds=false --out $O/tmp.json 2>&1 | grep "sim prof" > $O/sim_prof_c5_no_rounds.txt
# host budget of one rank of eight on a 16-CPU quota (zero_num_threads=1) against the default, incl. 450-move runs of C5 that cross two sequence boundaries (OBS compression)
timeout 600 python tools/run_configs.py c2 c3 c4 c5 --threads 1 --out $O/configs_threads1.json > $O/configs_threads1.log 2>&1
timeout 300 python tools/run_configs.py c5 --moves 450 --threads 1 --out $O/c5_450moves_threads1.json > /dev/null 2>&1
timeout 300 python tools/run_configs.py c5 --moves 450 --out $O/c5_450moves.json > /dev/null 2>&1
timeout 300 python tools/run_configs.py c5 --conf mz_sim_rounds=false --out $O/c5_no_rounds.json > /dev/null 2>&1
# round 4: what each piece of the rounds' evaluation is worth on the same box (one workgroup per leaf everywhere = round 3's path; no pairs; no adaptive second leaves)
timeout 300 python tools/run_configs.py c5 --conf mz_sim_round_batch=false:mz_sim_round_pairs=false --out $O/c5_round3_path.json > /dev/null 2>&1
timeout 300 python tools/run_configs.py c5 --conf mz_sim_round_pairs=false --out $O/c5_no_pairs.json > /dev/null 2>&1
timeout 600 python tools/run_configs.py c2 c3 c4 c5 --conf mz_rng_streams=1 --out $O/configs_one_rng_stream.json > /dev/null 2>&1
# kernel timeline of one C5 move (which launches, how long, the gaps between them)
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_c5 -- python tools/run_configs.py c5 --moves 20 --out $O/tmp.json > /dev/null 2>&1
python tools/c5_move_timeline.py $(find $O/trace_c5 -name "*kernel_trace.csv" | head -1) > $O/c5_move_timeline.txt 2>&1
rm -rf $O/trace_c5
rm -rf $O/stats $O/stats_c* $O/pmc1a $O/pmc1b $O/pmc2 $O/pmcA_* $O/pmcB_* $O/pmcC_* $O/tmp.json
ls -la $O | head -40; cat $O/pmc_sim.json; for k in c3 c4 c5 w9x128 w19x64; do cat $O/pmc_$k.json; done; python tools/kstats.py $O/kernel_stats.csv | head -6
python - <<'PY'
import json
j = json.load(open("gpurun_out/refresh/bench.json")); print("f32", round(j["value"]), j["ms_per_step"], j["games_per_sec"], j["roofline"]["frac"], j["cpu_baseline"]["value"])
j = json.load(open("gpurun_out/refresh/bench_bf16x3.json")); print("bf16x3", round(j["value"]), j["ms_per_step"], j["games_per_sec"], j["roofline"]["frac"])
d = json.load(open("gpurun_out/refresh/configs.json"))
for k, v in d.items(): print(k, round(v["leaf_evals_per_sec"]), v["roofline"]["frac"], v["roofline"]["wall_frac"])
PY
