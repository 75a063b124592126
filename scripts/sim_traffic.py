#!/usr/bin/env python3
"""Compulsory device-memory traffic of every launch of a forward pass, from the functional simulator's footprint build
(make -C sim traffic): the 64-byte lines each launch reads / writes at least once, and the bytes it requests.  Printed per
operator next to the HBM-side traffic rocprofv3 MEASURED for that operator in round 3 (profiles/r03_*_per_op_counters.txt),
so that "measured / compulsory" separates what the kernel decomposition makes necessary from what the caches failed to
keep.  CPU only; a model of nothing — it counts.

usage: sim_traffic.py <workload> [scans=4]        workload: ssv2_64x2048 | darknet53_64x2048 | darknet21_32x1024"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL = {"ssv2_64x2048": ("squeezesegv2", "squeezesegv2kitti", 64, 2048, 0.78), "darknet53_64x2048": ("darknet53", "darknet53kitti", 64, 2048, 0.78),
      "darknet21_32x1024": ("darknet21", "darknet21", 32, 1024, 0.59)}
CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import conftest
import pclsegmentation_amd as P
from pclsegmentation_amd import engine as E
from pclsegmentation_amd.utils.synthetic import synthetic_scans
name, cfg, h, w, pv, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
mc, model = P.load_model_config(name, cfg, height=h, width=w)
model.init_weights(4321)
model.micro_batch = n
raw = synthetic_scans(n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, pv, seed=1234)
import torch
scans = torch.from_numpy(raw)
preds = torch.empty((n, h, w), dtype=torch.int32)
eng = model.engine(h, w)
eng.forward_raw(scans, n, preds, None, None, None, mem=E.MEM_DEVICE)     # resident input, as bench.py times it
eng.sync()
open(sys.argv[7], "w").close()
eng.forward_raw(scans, n, preds, None, None, None, mem=E.MEM_DEVICE)
eng.sync()
print("\n".join(E.plan_ops(eng.desc)))
"""


def measured_r3(workload):
  path = os.path.join(ROOT, "profiles", "r03_%s_per_op_counters.txt" % workload)
  out = {}
  if os.path.exists(path):
    for line in open(path):
      f = line.split()
      if len(f) > 6 and re.match(r"^[\d.]+$", f[1]):
        out[f[0]] = (float(f[5]), float(f[6]))      # fetchMB, writeMB per 4-scan micro-batch
  return out


def main():
  wl, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
  name, cfg, h, w, pv = WL[wl]
  with tempfile.TemporaryDirectory() as tmp:
    log = os.path.join(tmp, "traffic.txt")
    env = dict(os.environ, PCLSEG_SIM="traffic", HIPSIM_TRAFFIC_LOG=log, PCLSEG_LANES="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, name, cfg, str(h), str(w), str(pv), str(n), log], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    if r.returncode:
      sys.exit(r.stdout[-2000:] + r.stderr[-4000:])
    ops = ["preprocess"] + [l for l in r.stdout.strip().splitlines() if l and not l.startswith("pclseg:")]
    rows = [l.rstrip("\n").split("\t") for l in open(log)]
  assert len(rows) == len(ops), (len(rows), len(ops))
  meas = measured_r3(wl) if n == 4 else {}
  print("# %s, %d scans in one micro-batch, one lane: unique 64-byte lines of device memory per launch (functional simulator, CPU) vs round 3's rocprofv3 PMC figures" % (wl, n))
  print("%-38s %9s %9s %9s %9s | %9s %9s %7s" % ("op", "read MB", "write MB", "req rd MB", "req wr MB", "r3 fetch", "r3 write", "meas/cmp"))
  tot = [0.0] * 6
  for op, (kern, grid, rd, wr, qr, qw) in zip(ops, rows):
    rd, wr, qr, qw = (int(x) / 1e6 for x in (rd, wr, qr, qw))
    m = meas.get(op)
    ratio = "%7.2f" % ((m[0] + m[1]) / (rd + wr)) if m and rd + wr > 0 else ""
    print("%-38s %9.1f %9.1f %9.1f %9.1f | %9s %9s %7s" % (op[:38], rd, wr, qr, qw, "%.1f" % m[0] if m else "", "%.1f" % m[1] if m else "", ratio))
    for i, v in enumerate((rd, wr, qr, qw)):
      tot[i] += v
    if m:
      tot[4] += m[0]; tot[5] += m[1]
  print("%-38s %9.1f %9.1f %9.1f %9.1f | %9.1f %9.1f %7s" % ("TOTAL per %d scans" % n, tot[0], tot[1], tot[2], tot[3], tot[4], tot[5],
        "%7.2f" % ((tot[4] + tot[5]) / (tot[0] + tot[1])) if tot[4] else ""))
  print("per scan: compulsory %.1f MB (read %.1f + write %.1f)" % ((tot[0] + tot[1]) / n, tot[0] / n, tot[1] / n))


if __name__ == "__main__":
  main()
