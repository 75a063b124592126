#!/bin/bash
# The GPU suite on the UndefinedBehaviorSanitizer build of the functional simulator (-fno-sanitize-recover: the first
# misaligned 16-byte vector access, out-of-range shift, signed overflow or out-of-bounds array index aborts the run).
# CPU only; about 25 minutes.  usage: scripts/sim_ubsan.sh [log]
cd "$(dirname "$0")/.."
log=${1:-/tmp/sim_ubsan.log}
make -s -C sim ubsan || exit 1
export LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
export UBSAN_OPTIONS=print_stacktrace=1
export PCLSEG_SIM=ubsan
{
  echo "# scripts/sim_ubsan.sh: GPU tests on sim/_build/libpclseg_sim_ubsan.so (UBSan: alignment, shift, signed-integer-overflow, null, bounds, float-cast-overflow; no recover), $(date -u +%F)"
  python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
  python -m pytest tests -m gpu -q -p no:cacheprovider -k "not full_size" --deselect tests/test_gpu_ops.py --deselect tests/test_sim_only.py 2>&1 | tail -1
} > "$log" 2>&1
cat "$log"
