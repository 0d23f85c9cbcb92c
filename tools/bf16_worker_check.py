"""bf16x3 in the worker: records vs the f32 mode over a few moves of the BASELINE Go config (expected: identical or nearly so — the outputs
differ by ~1e-6 — but NOT guaranteed), and throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz
key = sys.argv[1] if len(sys.argv) > 1 else "c2"
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = mz.DESCS[key]()
w = mz.generate_weights(d, 0)
n = int(mz.CONFIGS[key].split("actor_num_simulation=")[1].split(":")[0])
games = int(mz.CONFIGS[key].split("zero_num_parallel_games=")[1].split(":")[0])
out = {}
for prec in ("f32", "bf16x3"):
    wk = mz.Worker(f"{mz.CONFIGS[key]}:program_seed=1:nn_file_name=s.pt:zero_num_threads=8:mz_nn_precision={prec}", d, w)
    wk.command("start")
    wk.run_cycles(n + 1)
    t0 = time.perf_counter()
    for _ in range(moves):
        wk.run_cycles(n + 1)
    dt = time.perf_counter() - t0
    st = wk.stats()
    out[prec] = wk.peek_records(games)
    print(prec, "leaf-evals/s %.0f" % (games * (n + 1) * moves / dt), "sim_launches", st["sim_launches"], flush=True)
    del wk
same = sum(a == b for a, b in zip(out["f32"], out["bf16x3"]))
print(f"games with identical records after {moves + 1} moves: {same} of {games}")
