#!/bin/bash
# Same-box A/B of two builds of libmzgpu.so (ab/old.so, ab/new.so — not tracked): alternates them over `tools/run_configs.py <args>`.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p ab && cp minizero_amd/libmzgpu.so ab/keep.so
for i in 1 2 3; do
  for v in old new; do
    cp ab/$v.so minizero_amd/libmzgpu.so
    echo -n "$v: "; python tools/run_configs.py "$@" --out gpurun_out/tmp.json 2>&1 | grep -E "^c[0-9]" | cut -c1-75
  done
done
cp ab/keep.so minizero_amd/libmzgpu.so
