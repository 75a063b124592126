#!/bin/bash
# per-operator SQ counters (one lane): where the waves spend their time.  usage: pmc_sq.sh [ENV=V ...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PCLSEG_LANES=1 "$@"
B="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary"
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/kt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq -- $B > gpurun_out/pmc_sq.log 2>&1
python3 profiles/per_op_breakdown.py $(find gpurun_out/kt -name '*kernel_trace.csv' | head -1) > gpurun_out/kt.txt
python3 - <<'PY'
import csv, glob, collections, statistics
f = glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pclseg" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
  by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
names = [l.split()[1] for l in open("gpurun_out/kt.txt").read().splitlines() if len(l.split()) > 3 and l.split()[3] == "us"]
per = len(names)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i, d in enumerate(ids):
  for k, v in by[d].items():
    agg[i % per][k].append(v)
print("%-16s %9s %6s %6s %6s %6s %6s" % ("op", "wavecyc", "wait%", "istall%", "active%", "valu%", "mfma_busy/busy"))
for i in range(per):
  g = {k: statistics.median(v) for k, v in agg[i].items()}
  wc = g.get("SQ_WAVE_CYCLES", 1) or 1
  print("%-16s %9.0f %6.1f %6.1f %6.1f %6.1f %6.3f" % (names[i], wc, 100 * g.get("SQ_WAIT_ANY", 0) / wc,
        100 * g.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * g.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * g.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(g.get("SQ_BUSY_CYCLES", 1), 1)))
PY
cat gpurun_out/kt.txt
