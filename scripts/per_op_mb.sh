#!/bin/bash
# per-operator time per SCAN at several micro-batch sizes (one lane)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for mb in "$@"; do
  rm -rf gpurun_out/mb_$mb
  ( export PCLSEG_LANES=1; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mb_$mb -- python3 bench.py --steps 4 --warmup 2 --cpu-seconds 0 --batch 32 --micro-batch $mb > /dev/null 2>&1 )
  python3 profiles/per_op_breakdown.py $(find gpurun_out/mb_$mb -name '*kernel_trace.csv' | head -1) > gpurun_out/mb_$mb.txt
done
python3 - "$@" <<'PY'
import sys
mbs = sys.argv[1:]
cols = [open("gpurun_out/mb_%s.txt" % m).read().splitlines() for m in mbs]
print("us per scan at micro-batch:", mbs)
tot = [0.0] * len(mbs)
for k in range(len(cols[0])):
  parts = cols[0][k].split()
  if len(parts) > 3 and parts[3] == "us":
    vals = [float(c[k].split()[2]) / int(m) for c, m in zip(cols, mbs)]
    tot = [a + b for a, b in zip(tot, vals)]
    print("%-18s" % parts[1], "  ".join("%7.2f" % v for v in vals))
print("%-18s" % "total", "  ".join("%7.1f" % v for v in tot))
PY
