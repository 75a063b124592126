#!/usr/bin/env python3
"""HBM-side traffic per scan from two rocprofv3 PMC passes of `bench.py` (FETCH_SIZE, WRITE_SIZE in
separate runs, kernel-trace only), summed over the engine's kernels.

usage: make_traffic_json.py <fetch_dir> <write_dir> <scans_profiled> > rNN_traffic.json
The file records the sha of the kernel sources it was measured on (bench.csrc_sha); bench.py only
quotes the figure while that sha matches the build it is running.
FETCH_SIZE is doubled (gfx950 counts a 128-byte request as 64 B, MI355X_MICROARCH.md HBM section)."""
import csv
import glob
import json
import os
import sys


def total(d, counter):
  f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)  # newest pass
  return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f))
             if r["Counter_Name"] == counter and "pclseg" in r["Kernel_Name"])


fetch_kb, write_kb = total(sys.argv[1], "FETCH_SIZE"), total(sys.argv[2], "WRITE_SIZE")
scans = int(sys.argv[3])
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (csrc_sha only)
print(json.dumps({
  "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary",
  "scans_profiled": scans,
  "FETCH_SIZE_KB_total": fetch_kb,
  "WRITE_SIZE_KB_total": write_kb,
  "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
  "hbm_bytes_per_scan": (2 * fetch_kb + write_kb) * 1024 / scans,
  "alg_bytes_per_scan": 728367104,
  "csrc_sha": bench.csrc_sha(),
}, indent=1))
