#!/usr/bin/env python3
"""A BASELINE config at full size (default c2: 256 games x 400 simulations, 32 883-node pools; usage: moves [c1..c5]): whole games in the default mode (per-game simulation
kernel, device rules, path speculation) and in the lock-step mode with the host engine must give identical records.  ~2 minutes of GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz  # noqa: E402

key = sys.argv[2] if len(sys.argv) > 2 else "c2"
moves = int(sys.argv[1]) if len(sys.argv) > 1 else 164
d = mz.DESCS[key]()
w = mz.generate_weights(d, 0)
threads = max(1, mz.usable_cpus() - 1)
out = {}
n = int(mz.CONFIGS[key].split("actor_num_simulation=")[1].split(":")[0])
for name, extra in (("sim", ""), ("lockstep_host", ":mz_device_env=false:mz_sim_kernel=false")):
    wk = mz.Worker(f"{mz.CONFIGS[key]}:program_seed=1:nn_file_name=synthetic.pt:zero_num_threads={threads}{extra}", d, w)
    wk.command("start")
    t0 = time.perf_counter()
    wk.run_cycles((n + 1) * moves)
    out[name] = wk.pop_lines()
    print(name, "records", len(out[name]), "seconds %.1f" % (time.perf_counter() - t0), flush=True)
    del wk
same = out["sim"] == out["lockstep_host"]
print("identical records:", same, "(", len(out["sim"]), "games )")
sys.exit(0 if same and len(out["sim"]) > 0 else 1)
