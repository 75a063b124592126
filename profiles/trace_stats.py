#!/usr/bin/env python3
"""Per-kernel statistics (the columns of rocprofv3 --stats: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs,
MaxNs, StdDev) aggregated from a rocprofv3 --kernel-trace CSV, for passes that were run without --stats.
usage: trace_stats.py <kernel_trace.csv> > stats.csv"""
import collections
import csv
import statistics
import sys

by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
  by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in by.values())
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
  w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 6), round(100.0 * sum(v) / tot, 2), min(v), max(v),
              round(statistics.pstdev(v), 6)])
