#!/bin/bash
# per-operator A/B (one lane) of tuning knobs: usage ab_prof.sh "K=V K=V" "K=V" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/ab_$i
  ( export PCLSEG_LANES=1; [ -n "$cfg" ] && export $cfg; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$i -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-secondary > /dev/null 2>&1 )
  f=$(find gpurun_out/ab_$i -name '*kernel_trace.csv' | head -1)
  python3 profiles/per_op_breakdown.py $f > gpurun_out/ab_$i.txt
done
python3 - "$@" <<'PY'
import sys
cfgs = sys.argv[1:]
cols = []
for i in range(len(cfgs)):
  d, order = {}, []
  for line in open("gpurun_out/ab_%d.txt" % (i + 1)).read().splitlines():
    parts = line.split()
    if len(parts) > 3 and parts[3] == "us":
      d[parts[1]] = parts[2]; order.append(parts[1])
    elif line.startswith("sum"):
      d["__sum__"] = line[:60]
  cols.append((d, order))
print("configs:", cfgs)
names = []
for d, order in cols:
  for n in order:
    if n not in names: names.append(n)
for n in names:
  print("%-20s" % n, "  ".join("%8s" % d.get(n, "-") for d, _ in cols))
for d, _ in cols: print(d.get("__sum__", ""))
PY
