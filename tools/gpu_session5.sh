#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s5; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc $?"
timeout 600 python bench.py --steps 20 --warmup 5 --precision bf16x3 --no-cpu-baseline > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err; echo "bench bf16 rc $?"
python - <<'PY'
import json
for n in ("f32","bf16x3"):
    j=json.load(open(f"gpurun_out/s5/bench_{n}.json"))
    print(n, round(j["value"]), "ms/step %.2f"%j["ms_per_step"], "games/s %.1f"%j["games_per_sec"], "frac %.3f"%j["roofline"]["frac"], "ach %.1f"%j["roofline"]["achieved"], j["dtype"][:20])
PY
MZ_SIM_PROF=1 timeout 300 python bench.py --steps 4 --warmup 1 --game-moves 0 --no-cpu-baseline --precision bf16x3 2>&1 | grep "mz sim prof" | tee $O/sim_prof_bf16x3.txt
MZ_SIM_PROF=1 timeout 300 python bench.py --steps 4 --warmup 1 --game-moves 0 --no-cpu-baseline 2>&1 | grep "mz sim prof" | tee $O/sim_prof_f32.txt
