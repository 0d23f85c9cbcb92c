cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 1 2 3 4; do MZ_REPR_PARTS=$k python tools/time_atari_root.py 64 3; done
for i in 1 2 3; do
  for k in 1 2 3 4; do echo -n "parts $k: "; MZ_REPR_PARTS=$k python tools/run_configs.py c5 --out gpurun_out/tmp.json 2>&1 | grep -E "^c5" | cut -c1-75; done
done
