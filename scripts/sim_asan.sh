#!/bin/bash
# The operator suite (with its fuzz cases) and a network slice on the AddressSanitizer build of the functional simulator:
# every global / LDS / host access of the kernels and of the host code is checked (GPU sanitizers are not offered on
# this pool).  CPU only; about 10 minutes.  usage: scripts/sim_asan.sh [log]
cd "$(dirname "$0")/.."
log=${1:-/tmp/sim_asan.log}
make -s -C sim asan || exit 1
export LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1
export PCLSEG_SIM=asan
{
  echo "# scripts/sim_asan.sh: GPU tests on sim/_build/libpclseg_sim_asan.so (AddressSanitizer), $(date -u +%F)"
  python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
  python -m pytest tests/test_gpu_models.py tests/test_gpu_eval.py tests/test_gpu_projection.py tests/test_preproc_golden.py -m gpu -q -p no:cacheprovider \
    -k "(golden and f16x3 and (ssv2_32x240 or darknet53kitti)) or (golden and f32 and ssv2_32x240) or fully_fused or nan_pixel or range_fallback or confusion or projection or preprocessing" 2>&1 | tail -3
} > "$log" 2>&1
cat "$log"
