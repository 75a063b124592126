#!/usr/bin/env python3
"""Per-operator table of a one-lane profile made by scripts/per_op_profile.sh.

usage: per_op_table.py <workload> <dir with kt/ fetch/ write/ sq1/ sq2/ grbm/>
Launches are folded modulo the plan's launch sequence (names and MACs from pclseg_plan_ops, +1 for the
pre-processing launch); medians per position.  Columns:
  us        kernel duration (rocprofv3 kernel trace), micro-batch of `mb` scans
  TF16      f16 MFMA TFLOP/s executed = 3 products x 2 x MACs x mb / us   (exact-f32 layers: none)
  mfma%     that rate / 2500 TFLOP/s (dense f16 peak)
  busy%     SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   (counter view of the same)
  fetch/write MB  HBM-side bytes (FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE as is), TB/s = their sum / us
  wait% stall% active%  SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES;  ldsw% = SQ_WAIT_INST_LDS
  valu% = SQ_ACTIVE_INST_VALU;  conf% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;  clk = GRBM_GUI_ACTIVE / 8 / us (GHz)"""
import collections
import csv
import glob
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (WORKLOADS)
import pclsegmentation_amd as P  # noqa: E402
from pclsegmentation_amd import engine as E  # noqa: E402

wl, d = sys.argv[1], sys.argv[2]
model_name, config_name, h, w, batch, _, _ = bench.WORKLOADS[wl]
mc, model = P.load_model_config(model_name, config_name, height=h, width=w)
desc = model.engine_desc(h, w)
ops = [("preprocess", 0)] + E.plan_op_macs(desc)
mb = E.plan(desc)["micro_batch"]
per = len(ops)


def trace_rows(sub):
  f = glob.glob(os.path.join(d, sub, "**", "*kernel_trace.csv"), recursive=True)
  rows = [r for r in csv.DictReader(open(f[0])) if "pclseg" in r["Kernel_Name"]]
  rows.sort(key=lambda r: int(r["Start_Timestamp"]))
  return rows


def counters(sub):
  f = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
  if not f:
    return {}
  by = collections.defaultdict(dict)
  for r in csv.DictReader(open(f[0])):
    if "pclseg" in r["Kernel_Name"]:
      by[int(r["Dispatch_Id"])][r["Counter_Name"]] = by[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for i, k in enumerate(sorted(by)):
    for c, v in by[k].items():
      agg[i % per][c].append(v)
  return {i: {c: statistics.median(v) for c, v in cs.items()} for i, cs in agg.items()}


rows = trace_rows("kt")
if len(rows) % per:
  print("# warning: %d launches are not a multiple of the plan's %d" % (len(rows), per))
dur = collections.defaultdict(list)
for i, r in enumerate(rows):
  dur[i % per].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
fe, wr, s1, s2, gr = (counters(x) for x in ("fetch", "write", "sq1", "sq2", "grbm"))
print("# %s, one lane, micro-batch %d scans, %d launches per micro-batch; medians over %d micro-batches" % (wl, mb, per, len(rows) // per))
print("%-34s %7s %6s %6s %6s %8s %8s %6s %5s %6s %7s %5s %5s %5s %5s" % (
  "op", "us", "TF16", "mfma%", "busy%", "fetchMB", "writeMB", "TB/s", "wait%", "stall%", "active%", "ldsw%", "valu%", "conf%", "clk"))
tot = collections.defaultdict(float)
for i, (name, macs) in enumerate(ops):
  us = statistics.median(dur[i]) if dur[i] else 0.0
  tf = 3 * 2 * macs * mb / us / 1e6 if us else 0.0
  f = fe.get(i, {}).get("FETCH_SIZE", 0.0) * 2 * 1024 / 1e6
  wv = wr.get(i, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
  g1, g2 = s1.get(i, {}), s2.get(i, {})
  wc = g1.get("SQ_WAVE_CYCLES", 0) or 1
  gui = gr.get(i, {}).get("GRBM_GUI_ACTIVE", 0.0)
  clk = gui / 8 / us / 1e3 if us else 0.0
  busy = 100 * g1.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024) if gui else 0.0
  conf = 100 * g2.get("SQ_LDS_BANK_CONFLICT", 0) / g2["SQ_LDS_IDX_ACTIVE"] if g2.get("SQ_LDS_IDX_ACTIVE") else 0.0
  r = rows[i] if i < len(rows) else None
  print("%-34s %7.1f %6.0f %6.1f %6.1f %8.1f %8.1f %6.2f %5.1f %6.1f %7.1f %5.1f %5.1f %5.1f %5.2f  %s" % (
    name[:34], us, tf, 100 * tf / 2500, busy, f, wv, (f + wv) / us if us else 0, 100 * g1.get("SQ_WAIT_ANY", 0) / wc,
    100 * g1.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * g1.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * g1.get("SQ_WAIT_INST_LDS", 0) / wc,
    100 * g1.get("SQ_ACTIVE_INST_VALU", 0) / wc, conf, clk,
    ("grid=%d vgpr=%s %s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["VGPR_Count"], r["Kernel_Name"][8:58])) if r else ""))
  tot["us"] += us; tot["macs"] += macs; tot["f"] += f; tot["w"] += wv
print("total per micro-batch: %.1f us, %.1f f16 TFLOP/s executed (%.1f %% of 2500), fetch %.1f MB + write %.1f MB = %.1f MB per scan (%.2f TB/s)" % (
  tot["us"], 3 * 2 * tot["macs"] * mb / tot["us"] / 1e6, 100 * 3 * 2 * tot["macs"] * mb / tot["us"] / 1e6 / 2500,
  tot["f"], tot["w"], (tot["f"] + tot["w"]) / mb, (tot["f"] + tot["w"]) / tot["us"]))
