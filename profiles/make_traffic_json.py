#!/usr/bin/env python3
"""HBM-side traffic per scan of every workload from two rocprofv3 PMC passes of `bench.py` each (FETCH_SIZE,
WRITE_SIZE in separate runs, kernel-trace only), summed over the engine's kernels.

usage: make_traffic_json.py <round dir with pmc_fetch_<workload>/ and pmc_write_<workload>/> > rNN_traffic.json
The file records the sha of the kernel sources it was measured on (bench.csrc_sha); bench.py only quotes the
figures while that sha matches the build it is running.
FETCH_SIZE is doubled (gfx950 counts a 128-byte request as 64 B, MI355X_MICROARCH.md HBM section)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_sha, WORKLOADS)
import pclsegmentation_amd as P  # noqa: E402
from pclsegmentation_amd import engine as E  # noqa: E402


def totals(d, counter):
  """(sum of the counter over the engine's kernels, scans profiled = pre-processing launches x micro-batch)"""
  f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)  # newest pass
  tot, pre = 0.0, 0
  for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != counter or "pclseg" not in r["Kernel_Name"]:
      continue
    tot += float(r["Counter_Value"])
    pre += ("normalize_kernel" in r["Kernel_Name"]) and 1 or 0
  return tot, pre


rd = sys.argv[1]
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --workload <w> "
                  "--steps 5|3 --warmup 2 --cpu-seconds 0 --no-secondary",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
       "csrc_sha": bench.csrc_sha(), "binary_sha": E.build_sha(), "workloads": {}}
for wl in ("ssv2_64x2048", "darknet53_64x2048", "darknet21_32x1024"):
  fd, wd = os.path.join(rd, "pmc_fetch_" + wl), os.path.join(rd, "pmc_write_" + wl)
  if not (os.path.isdir(fd) and os.path.isdir(wd)):
    continue
  model_name, config_name, h, w, batch, _, _ = bench.WORKLOADS[wl]
  mc, model = P.load_model_config(model_name, config_name, height=h, width=w)
  info = E.plan(model.engine_desc(h, w))
  (fetch_kb, nf), (write_kb, nw) = totals(fd, "FETCH_SIZE"), totals(wd, "WRITE_SIZE")
  # every pre-processing launch covers one micro-batch; a batch of `batch` scans is dealt evenly over the micro-batches
  steps = 7 if wl != "darknet53_64x2048" else 5          # timed + warm-up steps of the command above
  scans = batch * steps
  out["workloads"][wl] = {
    "scans_profiled": scans, "preprocess_launches": [nf, nw], "FETCH_SIZE_KB_total": fetch_kb, "WRITE_SIZE_KB_total": write_kb,
    "hbm_bytes_per_scan": (2 * fetch_kb + write_kb) * 1024 / scans, "alg_bytes_per_scan": info["alg_bytes_per_scan"]}
print(json.dumps(out, indent=1))
