#!/bin/bash
# per-operator A/B (one lane) of tuning knobs: usage ab_prof.sh "K=V K=V" "K=V" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/ab_$i
  ( export PCLSEG_LANES=1 $cfg; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$i -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-secondary > /dev/null 2>&1 )
  f=$(find gpurun_out/ab_$i -name '*kernel_trace.csv' | head -1)
  python3 profiles/per_op_breakdown.py $f > gpurun_out/ab_$i.txt
done
python3 - "$@" <<'PY'
import sys
cfgs = sys.argv[1:]
cols = [open("gpurun_out/ab_%d.txt" % (i + 1)).read().splitlines() for i in range(len(cfgs))]
print("configs:", cfgs)
for k in range(len(cols[0])):
  parts = cols[0][k].split()
  if len(parts) > 3 and parts[3] == "us":
    print("%-18s" % parts[1], "  ".join("%8s" % c[k].split()[2] for c in cols))
  else:
    for c in cols: print(c[k][:60])
PY
