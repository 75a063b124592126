#!/usr/bin/env python3
"""Per-operator breakdown of a rocprofv3 --kernel-trace CSV of `bench.py` (SqueezeSegV2 workload).

usage: per_op_breakdown.py <kernel_trace.csv>
The engine launches a fixed kernel sequence per micro-batch; launches are folded modulo that
sequence and the median duration of each position is printed."""
import collections
import csv
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pclseg" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def fire(p, up=False):
  return [p + "/squeeze"] + ([p + "/upconv"] if up else []) + [p + "/expand"]


ops = ["normalize", "conv1", "cam1", "pool1"] + fire("fire2") + ["cam2"] + fire("fire3") + ["cam3", "pool3"]
ops += fire("fire4") + fire("fire5") + ["pool5"]
for f in ("fire6", "fire7", "fire8", "fire9"):
  ops += fire(f)
for f in ("fire10", "fire11", "fire12", "fire13"):
  ops += fire(f, True)
ops += ["conv14+head"]
per = int(sys.argv[2]) if len(sys.argv) > 2 else len(ops)
agg = collections.defaultdict(list)
for i, r in enumerate(rows):
  agg[i % per].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for i in range(per):
  m = statistics.median(agg[i])
  tot += m
  nm = ops[i] if i < len(ops) else "?"
  r = rows[i]
  print("%2d %-16s %8.1f us  grid=%dx%s lds=%s vgpr=%s+%s %s" % (
    i, nm, m / 1e3, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"],
    r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Kernel_Name"][8:44]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("sum of medians per micro-batch: %.1f us; %d launches; busy %.1f%% of the traced span"
      % (tot / 1e3, len(rows), 100.0 * busy / span))
