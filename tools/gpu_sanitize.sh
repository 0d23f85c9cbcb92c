#!/bin/bash
# The worker's host half under AddressSanitizer / ThreadSanitizer ON THE GPU (the real device behind it: the simulation-kernel paths of runCyclesSim, the Gumbel rounds,
# the OBS compressor with real Atari records — what the GPU-less harness of tests/test_sanitizers.py cannot reach).  The sanitizer builds of the library
# (make -C minizero_amd/csrc SAN=address ../libmzgpu_asan.so; SAN=thread ../libmzgpu_tsan.so: host sources instrumented, HIP objects as they are) are loaded by the
# ctypes mirror through MZ_LIBMZGPU; python itself is not instrumented, so the runtime is preloaded.  Output: gpurun_out/sanitize/{asan,tsan}.log
#   usage (through gpurun): bash tools/gpu_sanitize.sh [asan|tsan|both]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/sanitize
mkdir -p $O
# (the sanitizer builds stay out of the gpurun snapshot, .gpurunignore: 32 MB; the box has the same toolchain and builds them in a minute)
[ -f minizero_amd/libmzgpu_asan.so ] || make -s -j8 -C minizero_amd/csrc SAN=address ../libmzgpu_asan.so
[ -f minizero_amd/libmzgpu_tsan.so ] || make -s -j8 -C minizero_amd/csrc SAN=thread ../libmzgpu_tsan.so
WHAT=${1:-both}
TESTS_ASAN="tests/test_gpu_worker.py tests/test_gpu_streams.py tests/test_gpu_baseline_nets.py tests/test_gpu_wide_worker.py tests/test_gpu_loader.py tests/test_gpu_facade.py"
TESTS_TSAN="tests/test_gpu_worker.py tests/test_gpu_iteration.py" # (not test_gpu_streams.py: its CHECKER, the uninstrumented oracle with T threads of its own, does not survive under the preloaded runtime)
if [ "$WHAT" = asan ] || [ "$WHAT" = both ]; then
  ( time env LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:exitcode=66 \
      MZ_LIBMZGPU=$PWD/minizero_amd/libmzgpu_asan.so timeout 1500 python -m pytest $TESTS_ASAN tests/test_gpu_iteration.py -q -m gpu -k "not executable and not sp_exe" -x 2>&1 | tail -40 ) > $O/asan.log 2>&1
  tail -5 $O/asan.log
fi
if [ "$WHAT" = tsan ] || [ "$WHAT" = both ]; then
  # (ThreadSanitizer and the ROCm runtime both want large fixed address ranges: if the runtime refuses to start under TSAN the log says so — recorded, not hidden)
  # (setarch -R: gcc 11's ThreadSanitizer refuses the address-space randomisation of newer kernels — "unexpected memory mapping" — before python even starts)
  ( time setarch x86_64 -R env LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS=exitcode=66:halt_on_error=0:report_signal_unsafe=0:ignore_noninstrumented_modules=1 \
      MZ_LIBMZGPU=$PWD/minizero_amd/libmzgpu_tsan.so timeout 900 python -m pytest $TESTS_TSAN -q -m gpu -k "c1 or small_go or multithreaded or atari_gumbel or commands or muzero or (games_in_flight and not c5_net and not c2_net) or reset_actors" --deselect "tests/test_gpu_iteration.py::test_iteration_from_a_real_torchscript_file" 2>&1 | tail -60 ) > $O/tsan.log 2>&1
  tail -8 $O/tsan.log
fi
