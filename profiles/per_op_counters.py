#!/usr/bin/env python3
"""Per-operator HBM-side traffic and SQ counters of a one-lane run, folded like per_op_breakdown.py.

usage: per_op_counters.py <per_op.txt> <fetch_dir> <write_dir> <sq_dir>
FETCH_SIZE is doubled (gfx950 counts a 128-byte request as 64 B, MI355X_MICROARCH.md HBM section)."""
import collections
import csv
import glob
import statistics
import sys

names = [l.split()[1] for l in open(sys.argv[1]).read().splitlines() if len(l.split()) > 3 and l.split()[3] == "us"]
times = [float(l.split()[2]) for l in open(sys.argv[1]).read().splitlines() if len(l.split()) > 3 and l.split()[3] == "us"]
per = len(names)


def load(d):
  f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
  if not f:
    return {}
  by = collections.defaultdict(dict)
  for r in csv.DictReader(open(f[0])):
    if "pclseg" in r["Kernel_Name"]:
      by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for i, d_ in enumerate(sorted(by)):
    for k, v in by[d_].items():
      agg[i % per][k].append(v)
  return {i: {k: statistics.median(v) for k, v in c.items()} for i, c in agg.items()}


fe, wr, sq = load(sys.argv[2]), load(sys.argv[3]), load(sys.argv[4])
print("%-20s %8s %9s %9s %8s %6s %7s %7s %6s" % ("op", "us", "fetch MB", "write MB", "TB/s", "wait%", "istall%", "active%", "valu%"))
tf = tw = tt = 0.0
for i in range(per):
  f = fe.get(i, {}).get("FETCH_SIZE", 0.0) * 2 * 1024 / 1e6
  w = wr.get(i, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
  g = sq.get(i, {})
  wc = g.get("SQ_WAVE_CYCLES", 0) or 1
  tf += f; tw += w; tt += times[i]
  print("%-20s %8.1f %9.1f %9.1f %8.2f %6.1f %7.1f %7.1f %6.1f" % (
    names[i], times[i], f, w, (f + w) / times[i] if times[i] else 0, 100 * g.get("SQ_WAIT_ANY", 0) / wc,
    100 * g.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * g.get("SQ_ACTIVE_INST_ANY", 0) / wc,
    100 * g.get("SQ_ACTIVE_INST_VALU", 0) / wc))
print("total per micro-batch: %.1f us, fetch %.1f MB, write %.1f MB (%.2f TB/s)" % (tt, tf, tw, (tf + tw) / tt))
