#!/usr/bin/env python3
"""Per-operator breakdown of a rocprofv3 --kernel-trace CSV of `bench.py` (SqueezeSegV2 workload).

usage: per_op_breakdown.py <kernel_trace.csv> [launches per micro-batch]
The engine launches a fixed kernel sequence per micro-batch; launches are folded modulo that
sequence (its length is detected from the repeating kernel names) and the median duration of
each position is printed."""
import collections
import csv
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pclseg" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def fire(p, up=False, sq=True, fused=None):
  return ([p + "/squeeze"] if sq else []) + ([p + "/upconv"] if up else []) + \
         [p + "/expand" + ("+" + fused if fused else "")]


def sequence(fused, pooled=False, cam=False, up=True):
  """fused 0: every squeeze its own launch; 1: fireN's expand blocks also compute fireN+1's squeeze for
  N = 4, 6, 7, 8, 9; 2: also for the FIREUP chain N = 10, 11, 12.  pooled: pool1/3/5 run inside
  fire2/4/6's squeeze; cam: cam2's blocks compute fire3's squeeze; up=False: the FIREUP pairs
  up-convolve their own patch (no upconv launches)."""
  def psq(n, f):
    return ["pool%d+sq%d" % (n, n + 1)] + fire(f, sq=False) if pooled else ["pool%d" % n] + fire(f)
  ops = ["normalize", "conv1", "cam1"] + psq(1, "fire2")
  ops += (["cam2+sq3"] + fire("fire3", sq=False) if cam else ["cam2"] + fire("fire3")) + ["cam3"]
  if fused and pooled:
    ops += ["pool3+sq4"] + fire("fire4", sq=False, fused="sq5") + fire("fire5", sq=False)
    ops += ["pool5+sq6"] + fire("fire6", sq=False, fused="sq7")
  elif fused:
    ops += ["pool3"] + fire("fire4", fused="sq5") + fire("fire5", sq=False) + ["pool5"]
    ops += fire("fire6", fused="sq7")
  if fused:
    ops += fire("fire7", sq=False, fused="sq8") + fire("fire8", sq=False, fused="sq9")
    ops += fire("fire9", sq=False, fused="sq10")
    if fused == 2:
      ops += fire("fire10", up, sq=False, fused="sq11") + fire("fire11", up, sq=False, fused="sq12")
      ops += fire("fire12", up, sq=False, fused="sq13") + fire("fire13", up, sq=False)
    else:
      ops += fire("fire10", True, sq=False)
  else:
    ops += ["pool3"] + fire("fire4") + fire("fire5") + ["pool5"]
    for f in ("fire6", "fire7", "fire8", "fire9"):
      ops += fire(f)
    ops += fire("fire10", True)
  if fused != 2:
    for f in ("fire11", "fire12", "fire13"):
      ops += fire(f, True)
  return ops + ["conv14+head"]


names = [r["Kernel_Name"] for r in rows]
if len(sys.argv) > 2:
  per = int(sys.argv[2])
else:   # smallest period of the kernel-name sequence
  per = next((p for p in range(8, 80) if len(names) >= 3 * p and names[:2 * p] == names[p:3 * p]), 37)
ops = {37: sequence(0), 32: sequence(1), 29: sequence(2), 26: sequence(2, True), 25: sequence(2, True, True), 21: sequence(2, True, True, False)}.get(per, [])
if per == 20:   # round 3: fire13's expand pair, conv14 and the head are one launch
  ops = sequence(2, True, True, False)[:-2] + ["fire13+conv14+head"]
if not ops:     # Darknet (or an unknown plan): label by position + kernel family
  ops = []
agg = collections.defaultdict(list)
for i, r in enumerate(rows):
  agg[i % per].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for i in range(per):
  m = statistics.median(agg[i])
  tot += m
  nm = ops[i] if i < len(ops) else "op%d" % i
  r = rows[i]
  print("%2d %-18s %8.1f us  grid=%dx%s lds=%s vgpr=%s+%s %s" % (
    i, nm, m / 1e3, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"],
    r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Kernel_Name"][8:44]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("sum of medians per micro-batch: %.1f us; %d launches; busy %.1f%% of the traced span"
      % (tot / 1e3, len(rows), 100.0 * busy / span))
