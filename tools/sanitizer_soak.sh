#!/bin/bash
# Sanitizer run of the C-ABI host layer on the GPU box (sanitizer build):
#   make -C pygps_amd/csrc ubsan   then   gpurun -- 'bash tools/sanitizer_soak.sh'
# UndefinedBehaviorSanitizer with -fno-sanitize-recover: any finding aborts the process, so "tests passed" = clean.
# Exercises: set_data / exact fit with and without factor handles, non-PD error path, EP, FITC, predict, restart search
# with two fit streams from two host threads (pools + mutexes), handle release by finalizers, re-use after errors.
# (AddressSanitizer: `make asan` builds, but the image's HIP runtime is not ASan-enabled -- see csrc/Makefile.)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYGPS_AMD_LIB=$R/pygps_amd/libpygps_amd_ubsan.so
export UBSAN_OPTIONS=print_stacktrace=1
export LD_PRELOAD=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
cd $R
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_fitc.py tests/test_gpu_core.py tests/test_gpu_sharded.py tests/test_gpu_composite.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py tests/test_gpu_r5.py tests/test_gpu_r6.py -m gpu -x -q -k "not bench and not rccl and not world" 2>&1 | grep -E "passed|failed|error|runtime error" | tail -5
timeout 600 python tools/two_streams.py 2>&1 | tail -2
REPS=10 timeout 600 python tools/ep_kfold_diag.py 2>&1 | grep -E "rep|gave" | tail -3
timeout 600 python tools/mixed_soak.py 2>&1 | tail -3
