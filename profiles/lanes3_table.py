#!/usr/bin/env python3
"""The three-lane view of the headline workload, from the passes of scripts/lanes3_counters.sh.

usage: lanes3_table.py <dir with kt1/ kt3/ pmc1_<i>/ pmc3_<i>/>
A lane is a HIP stream (Stream_Id of the kernel trace); within a lane the launches follow the plan's
sequence (pclseg_plan_ops, +1 for the pre-processing launch), so launch k of a lane is operator k mod len(plan).

Section 0  static residency of every launch: LDS bytes and threads per block (pclseg_plan_ops), registers per lane
           (kernel trace), hence blocks per CU and what is left of a CU beside them — which kernels CAN share a CU.
Section 1  job level: kernel time per scan, wall time per scan, kernels in flight, at 1 and at 3 lanes.
Section 2  per operator: duration alone (1 lane) and among the other lanes' kernels (3 lanes), the inflation,
           the time-weighted number of OTHER kernels running beside it, and the operator it overlaps most.
Section 3  who slows whom: mean inflation of operator class A while class B is the main co-runner.
Section 4  per counter group: whole-job totals per scan at 1 and at 3 lanes and their ratio (a counter that counts
           WORK stays at 1.0; one that counts WAITING grows with contention), plus kernels in flight during that
           pass (rocprofv3 serialises dispatches while it collects counters: if this reads 1.0 the pass saw no
           co-residency and its 3-lane column is a second 1-lane measurement).
Section 5  per operator counter ratios 3 lanes / 1 lane for the wait / busy / occupancy-limiter counters."""
import collections
import csv
import glob
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pclsegmentation_amd as P  # noqa: E402
from pclsegmentation_amd import engine as E  # noqa: E402

d = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "ssv2_64x2048"
model_name, config_name, h, w, batch, _, _ = bench.WORKLOADS[wl]
mc, model = P.load_model_config(model_name, config_name, height=h, width=w)
desc = model.engine_desc(h, w)
res = [("preprocess", 0, 0, 256, 0)] + E.plan_op_resources(desc)
ops = [r[0] for r in res]
mb = E.plan(desc)["micro_batch"]
per = len(ops)


def klass(name):
  if name.startswith("cam"):
    return "cam"
  if name.startswith("pool") or "pool" in name.split("/")[0]:
    return "pool+sq"
  if "head" in name or name.startswith("fire13"):
    return "tail"
  if name in ("preprocess", "conv1"):
    return "stem"
  for k in ("fire2", "fire3", "fire4", "fire5"):
    if name.startswith(k):
      return "fire2-5"
  for k in ("fire6", "fire7", "fire8", "fire9"):
    if name.startswith(k):
      return "fire6-9"
  return "fire10-12"


def trace(sub):
  f = glob.glob(os.path.join(d, sub, "**", "*kernel_trace.csv"), recursive=True)
  if not f:
    return []
  # the bench process is the one with the most pclseg kernels
  best = []
  for path in f:
    rows = [r for r in csv.DictReader(open(path)) if "pclseg" in r["Kernel_Name"]]
    if len(rows) > len(best):
      best = rows
  lanes = collections.defaultdict(list)
  for r in best:
    lanes[r["Stream_Id"]].append(r)
  out = []
  for sid, rows in lanes.items():
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for k, r in enumerate(rows):
      out.append({"op": k % per, "lane": sid, "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"]),
                  "id": int(r["Dispatch_Id"]), "name": r["Kernel_Name"], "vgpr": int(r.get("VGPR_Count") or 0),
                  "blocks": int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))})
  out.sort(key=lambda r: r["t0"])
  return out


def job(rows):
  """(kernel us per scan, wall us per scan, kernels in flight) over the steady part of the trace."""
  if not rows:
    return 0.0, 0.0, 0.0
  t0, t1 = min(r["t0"] for r in rows), max(r["t1"] for r in rows)
  # busy wall = union of intervals (warm-up gaps between steps do not count)
  ev = sorted([(r["t0"], 1) for r in rows] + [(r["t1"], -1) for r in rows])
  busy, depth, last = 0, 0, ev[0][0]
  for t, s in ev:
    if depth > 0:
      busy += t - last
    depth += s
    last = t
  ksum = sum(r["t1"] - r["t0"] for r in rows)
  scans = len(rows) / per * mb
  return ksum / 1e3 / scans, busy / 1e3 / scans, ksum / max(busy, 1)


def overlaps(rows):
  """per row: time-weighted count of other kernels in flight, and overlap time per other operator."""
  import bisect
  starts = [r["t0"] for r in rows]
  res = []
  maxdur = max(r["t1"] - r["t0"] for r in rows)
  for i, r in enumerate(rows):
    lo = bisect.bisect_left(starts, r["t0"] - maxdur)
    hi = bisect.bisect_right(starts, r["t1"])
    tot, byop = 0, collections.Counter()
    for j in range(lo, hi):
      if j == i:
        continue
      o = rows[j]
      ov = min(r["t1"], o["t1"]) - max(r["t0"], o["t0"])
      if ov > 0:
        tot += ov
        byop[o["op"]] += ov
    res.append((tot / max(r["t1"] - r["t0"], 1), byop))
  return res


k1, k3 = trace("kt1"), trace("kt3")
print("# %s: micro-batch %d scans, %d launches per micro-batch (%s)" % (wl, mb, per, d))
print("\n## 0. static residency per launch (a CU: 160 KiB LDS, 4 SIMDs x 512 registers per lane x 8 wave slots)")
print("%-34s %7s %7s %5s %6s %9s %10s %14s" % ("op", "blocks", "LDS KB", "thr", "VGPRs", "blocks/CU", "limited by", "left on the CU"))
# registers per lane: the compiler's own figure (profiles/rNN_kernel_resources.txt from scripts/resources.sh),
# matched by kernel name; fallback 2 x the trace CSV's VGPR_Count (it counts register pairs on gfx950)
import re
regs = {}
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_resources.txt"))):
  for line in open(f):
    m = re.match(r"(.*?)\s+vgpr=\s*(\d+)", line)
    if m:
      regs[re.sub(r"\s+", "", m.group(1).replace("void ", ""))] = int(m.group(2))
def regs_of(kname, csv_count):
  key = re.sub(r"\s+", "", kname.replace("void ", "").replace("pclseg::", "").split("(")[0])
  for k, v in regs.items():
    if k.split("(")[0] == key:
      return -(-v // 8) * 8        # allocation granule: 8
  return 2 * csv_count
vg = {}
for r in k1:
  vg.setdefault(r["op"], (regs_of(r["name"], r["vgpr"]), r["blocks"]))
for i, (name, macs, lds, thr, bps) in enumerate(res):
  v, blocks = vg.get(i, (0, 0))
  wps = thr // 64 / 4.0                                    # waves per SIMD of one block
  by_lds = 163840 // lds if lds else 99
  by_vgpr = int((512 // max(v, 1)) // max(wps, 0.25)) if v else 99
  by_waves = int(8 // max(wps, 0.25))
  n = max(1, min(by_lds, by_vgpr, by_waves, 8))
  lim = "LDS" if n == by_lds else "registers" if n == by_vgpr else "wave slots"
  left = "%3d KB, %3d regs" % ((163840 - n * lds) // 1024, 512 - int(n * wps * v))
  print("%-34s %7d %7.1f %5d %6d %9d %10s %16s" % (name[:34], blocks, lds / 1024.0, thr, v, n, lim, left))
print("\n## 1. job level")
print("%-8s %14s %14s %10s" % ("lanes", "kernel us/scan", "wall us/scan", "in flight"))
j1, j3 = job(k1), job(k3)
print("%-8s %14.1f %14.1f %10.2f" % ("1", j1[0], j1[1], j1[2]))
print("%-8s %14.1f %14.1f %10.2f" % ("3", j3[0], j3[1], j3[2]))
if j1[1] and j3[1]:
  print("three lanes: wall x%.3f (throughput x%.3f), kernel time x%.2f" % (j3[1] / j1[1], j1[1] / j3[1], j3[0] / j1[0]))

d1 = collections.defaultdict(list)
for r in k1:
  d1[r["op"]].append((r["t1"] - r["t0"]) / 1e3)
d3 = collections.defaultdict(list)
co = collections.defaultdict(list)
partner = collections.defaultdict(collections.Counter)
ov3 = overlaps(k3) if k3 else []
for r, (n_other, byop) in zip(k3, ov3):
  d3[r["op"]].append((r["t1"] - r["t0"]) / 1e3)
  co[r["op"]].append(n_other)
  partner[r["op"]].update(byop)
print("\n## 2. per operator (medians)")
print("%-34s %8s %8s %7s %9s %12s  %s" % ("op", "1lane us", "3lane us", "x", "others", "extra us/mb", "overlaps most with"))
extra_tot = 0.0
for i, name in enumerate(ops):
  a = statistics.median(d1[i]) if d1[i] else 0.0
  b = statistics.median(d3[i]) if d3[i] else 0.0
  p = partner[i].most_common(1)
  extra_tot += b - a
  print("%-34s %8.1f %8.1f %7.2f %9.2f %12.1f  %s" % (name[:34], a, b, b / a if a else 0, statistics.mean(co[i]) if co[i] else 0,
                                                   b - a, ops[p[0][0]] if p else "-"))
print("%-34s %8.1f %8.1f %7.2f" % ("sum", sum(statistics.median(v) for v in d1.values() if v),
                                  sum(statistics.median(v) for v in d3.values() if v),
                                  sum(statistics.median(v) for v in d3.values() if v) / max(1e-9, sum(statistics.median(v) for v in d1.values() if v))))

print("\n## 3. who slows whom: mean inflation of class A (rows) while its main co-runner is of class B (columns)")
classes = ["stem", "cam", "pool+sq", "fire2-5", "fire6-9", "fire10-12", "tail"]
cell = collections.defaultdict(list)
med1 = {i: (statistics.median(v) if v else 0.0) for i, v in d1.items()}
for r, (n_other, byop) in zip(k3, ov3):
  if not byop or not med1.get(r["op"]):
    continue
  main = byop.most_common(1)[0][0]
  cell[(klass(ops[r["op"]]), klass(ops[main]))].append((r["t1"] - r["t0"]) / 1e3 / med1[r["op"]])
print("%-10s" % "A \\ B" + "".join("%10s" % c for c in classes))
for a in classes:
  print("%-10s" % a + "".join(("%10s" % ("%.2f (%d)" % (statistics.mean(cell[(a, b)]), len(cell[(a, b)])) if cell[(a, b)] else "%10s" % "-")) for b in classes))


def counter_pass(sub):
  f = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
  best = {}
  for path in f:
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
      if "pclseg" in r["Kernel_Name"]:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = by[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if len(by) > len(best):
      best = by
  return best


print("\n## 4. counter totals per scan, 1 lane vs 3 lanes (one pass per group and lane count)")
print("%-34s %16s %16s %8s   %s" % ("counter", "1 lane / scan", "3 lanes / scan", "ratio", "in flight during the 3-lane pass"))
groups = sorted({os.path.basename(p).split("_")[1] for p in glob.glob(os.path.join(d, "pmc3_*")) if os.path.isdir(p)}, key=int)
perop = {}
for g in groups:
  c1, c3 = counter_pass("pmc1_" + g), counter_pass("pmc3_" + g)
  t3 = trace("pmc3_" + g)
  infl = job(t3)[2] if t3 else 0.0
  names = sorted({c for v in c3.values() for c in v} | {c for v in c1.values() for c in v})
  if not names:
    log = os.path.join(d, "pmc3_%s.log" % g)
    msg = open(log).read().strip().splitlines()[-1][:120] if os.path.exists(log) and open(log).read().strip() else "no output"
    print("group %s: no counters collected (%s)" % (g, msg))
    continue
  for c in names:
    s1 = sum(v.get(c, 0.0) for v in c1.values()) / max(1, len(c1) / per * mb)
    s3 = sum(v.get(c, 0.0) for v in c3.values()) / max(1, len(c3) / per * mb)
    print("%-34s %16.4g %16.4g %8.3f   %.2f" % (c, s1, s3, s3 / s1 if s1 else 0.0, infl))
  # per-op medians for section 5: dispatch order within the pass follows launch order per lane only when serialised;
  # fold by kernel-trace lane of the same pass
  for tag, cc, tr in (("1", c1, trace("pmc1_" + g)), ("3", c3, t3)):
    opof = {r["id"]: r["op"] for r in tr}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for did, vals in cc.items():
      if did in opof:
        for c, v in vals.items():
          agg[opof[did]][c].append(v)
    for i, cs in agg.items():
      for c, v in cs.items():
        perop[(tag, i, c)] = statistics.median(v)

print("\n## 5. per operator, 3 lanes / 1 lane ratio of selected counters (medians per dispatch)")
sel = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_INST_LDS",
       "SPI_RA_RES_STALL_CSN", "SPI_RA_LDS_CU_FULL_CSN", "SPI_RA_VGPR_SIMD_FULL_CSN", "SPI_RA_WAVE_SIMD_FULL_CSN",
       "TCC_MISS_sum", "TCC_HIT_sum"]
sel = [c for c in sel if any(k[2] == c for k in perop)]
print("%-34s" % "op" + "".join("%12s" % c.replace("SQ_", "").replace("SPI_RA_", "")[:11] for c in sel))
for i, name in enumerate(ops):
  line = "%-34s" % name[:34]
  for c in sel:
    a, b = perop.get(("1", i, c)), perop.get(("3", i, c))
    line += "%12s" % ("%.2f" % (b / a) if a and b is not None else ("%.3g" % b if b else "-"))
  print(line)
