#!/bin/bash
# effective shader clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (MI355X_MICROARCH.md, DVFS)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/clk
export PCLSEG_LANES=1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk -- python3 bench.py --workload ${1:-darknet53_64x2048} --steps 6 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
d = "gpurun_out/clk"
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(cc)) if "pclseg" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
by = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in rows:
  dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
  k = r["Kernel_Name"][8:52] + " grid=%d" % (int(r["Grid_Size"]) // 256)
  by[k][0] += float(r["Counter_Value"]); by[k][1] += dur; by[k][2] += 1
tot_c = sum(v[0] for v in by.values()); tot_t = sum(v[1] for v in by.values())
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
  print("%-70s %6.0f us avg  clock %.2f GHz" % (k, v[1] / v[2] / 1e3, v[0] / 8 / v[1]))
print("all kernels: effective clock %.2f GHz" % (tot_c / 8 / tot_t))
PY
