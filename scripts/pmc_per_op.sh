#!/bin/bash
# per-operator L2-miss traffic (one lane): FETCH_SIZE and WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PCLSEG_LANES=1 $1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, statistics
def load(c):
  f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
  rows = [r for r in csv.DictReader(open(f)) if "pclseg" in r["Kernel_Name"] and r["Counter_Name"] == c]
  rows.sort(key=lambda r: int(r["Dispatch_Id"]))
  return rows
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
import sys
sys.path.insert(0, "profiles")
def fire(p, up=False):
  return [p + "/squeeze"] + ([p + "/upconv"] if up else []) + [p + "/expand"]
ops = ["normalize", "conv1", "cam1", "pool1"] + fire("fire2") + ["cam2"] + fire("fire3") + ["cam3", "pool3"]
ops += fire("fire4") + fire("fire5") + ["pool5"]
for f in ("fire6", "fire7", "fire8", "fire9"): ops += fire(f)
for f in ("fire10", "fire11", "fire12", "fire13"): ops += fire(f, True)
ops += ["conv14+head"]
per = len(ops)
a = collections.defaultdict(list); b = collections.defaultdict(list)
for i, r in enumerate(fe): a[i % per].append(float(r["Counter_Value"]) * 2 * 1024)   # KB, x2 gfx950 correction
for i, r in enumerate(wr): b[i % per].append(float(r["Counter_Value"]) * 1024)
tf = tw = 0
for i in range(per):
  f, w = statistics.median(a[i]), statistics.median(b[i])
  tf += f; tw += w
  print("%-16s fetch %7.1f MB  write %7.1f MB" % (ops[i], f / 1e6, w / 1e6))
print("total per micro-batch: fetch %.1f MB write %.1f MB" % (tf / 1e6, tw / 1e6))
PY
