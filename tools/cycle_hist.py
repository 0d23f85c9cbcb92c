#!/usr/bin/env python3
"""Per-cycle wall-time distribution of the C2 worker (diagnostic for host-side stalls): times run_cycles(1) N times for
several host-thread counts and prints percentiles, the outliers, and the cgroup CPU throttle counters before/after."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import minizero_amd as mz  # noqa: E402


def cg():
    out = {}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            out[p] = open(p).read().strip().replace("\n", " | ")
        except Exception:
            pass
    return out


print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "loadavg", os.getloadavg())
print(json.dumps(cg(), indent=1))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
for threads, extra in [(32, ""), (16, ""), (8, ""), (32, ":mz_cpu_base=-1"), (32, ":mz_signal_wait=false")]:
    d = mz.DESCS["c2"]()
    conf = f"{mz.CONFIGS['c2']}:program_seed=1:nn_file_name=s.pt:zero_num_threads={threads}:mz_cpu_base=0{extra}"
    wk = mz.Worker(conf, d, mz.generate_weights(d, 0))
    wk.command("start")
    wk.run_cycles(40)
    ts = np.empty(N)
    c0 = cg().get("/sys/fs/cgroup/cpu.stat", "")
    for i in range(N):
        t0 = time.perf_counter()
        wk.run_cycles(1)
        ts[i] = time.perf_counter() - t0
    ts *= 1e3
    big = np.where(ts > 5 * np.median(ts))[0]
    print(f"threads={threads}{extra}: mean {ts.mean():.3f} ms  median {np.median(ts):.3f}  p90 {np.percentile(ts, 90):.3f}  p99 {np.percentile(ts, 99):.3f}  "
          f"max {ts.max():.2f}  evals/s(mean) {256 / ts.mean() * 1e3:.0f}  evals/s(median) {256 / np.median(ts) * 1e3:.0f}  "
          f"outliers(>5x median) n={len(big)} sum={ts[big].sum():.1f} ms of {ts.sum():.1f}", flush=True)
    print("   outlier idx:", big[:20].tolist(), np.round(ts[big][:20], 1).tolist())
    print("   cpu.stat before:", c0, "\n   cpu.stat after: ", cg().get("/sys/fs/cgroup/cpu.stat", ""), flush=True)
    del wk
