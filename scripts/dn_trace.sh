#!/bin/bash
# kernel sequence of one Darknet-53 micro-batch (one lane): duration, grid, kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/dn
( export PCLSEG_LANES=1; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dn -- python3 bench.py --workload ${1:-darknet53_64x2048} --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary > /dev/null 2>&1 )
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dn/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pclseg" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one micro-batch = from one normalize kernel to the next
starts = [i for i, r in enumerate(rows) if "normalize" in r["Kernel_Name"]]
seg = rows[starts[-2]:starts[-1]]
tot = 0
for i, r in enumerate(seg):
  d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
  tot += d
  print("%3d %8.1f us grid=%6d %s" % (i, d, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Kernel_Name"][8:60]))
print("total %.1f us, %d launches" % (tot, len(seg)))
PY
