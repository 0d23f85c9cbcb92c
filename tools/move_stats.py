import sys, os, time, json
sys.path.insert(0, '/root/repo')
import minizero_amd as mz
key = sys.argv[1] if len(sys.argv) > 1 else 'c3'
d = mz.DESCS[key]()
threads = mz.usable_cpus() - 1
conf = f"{mz.CONFIGS[key]}:program_seed=1:nn_file_name=synthetic.pt:zero_num_threads={threads}"
wk = mz.Worker(conf, d, mz.generate_weights(d, 0))
wk.command("start")
n = int(mz.CONFIGS[key].split("actor_num_simulation=")[1].split(":")[0]) + 1
wk.run_cycles(n * 2)
s0 = wk.stats(); t0 = time.perf_counter()
wk.run_cycles(n * 30)
dt = time.perf_counter() - t0; s1 = wk.stats()
print("threads", threads, "ms/move", dt / 30 * 1e3, "evals/s", (s1["leaf_evals"] - s0["leaf_evals"]) / dt)
print({k: round((s1[k] - s0[k]) / 30, 3) for k in s1 if k.startswith("ms_")})
