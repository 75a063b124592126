#!/usr/bin/env python3
"""Scale a one-scan compulsory-traffic table of scripts/sim_traffic.py to the micro-batch round 3 MEASURED with, and put the
measured PMC figures beside it.  (Darknet-53 64x2048 takes 2 h 45 min per scan in the footprint build; four scans would not
finish in a session.)  Per launch: compulsory(n scans) = n x (read + write at one scan) - (n - 1) x W, W = the launch's weight
fragments, read once whatever the micro-batch: 4 bytes (hi + lo half) per folded weight of the layer's Keras kernel.

usage: sim_traffic_scale.py <table of `sim_traffic.py <workload> 1`> <workload> [scans=4]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import sim_traffic  # noqa: E402
import pclsegmentation_amd as P  # noqa: E402


def main():
  path, wl, n = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4
  name, cfg, h, w, _ = sim_traffic.WL[wl]
  _, model = P.load_model_config(name, cfg, height=h, width=w)
  wbytes = {}
  for ws in model.weight_spec():
    if ws.path.endswith("/kernel"):
      k = 1
      for d in ws.shape:
        k *= int(d)
      wbytes[ws.path[:-len("/kernel")]] = 4 * k
  meas = sim_traffic.measured_r3(wl)
  rows = []
  for line in open(path):
    f = line.split("|")[0].split()
    if len(f) == 5 and f[0] not in ("op", "TOTAL") and not line.startswith("#"):
      rows.append((f[0], float(f[1]), float(f[2])))
  print("# %s: compulsory device traffic per launch at %d scans per micro-batch, scaled from the one-scan count (%s), beside round 3's"
        % (wl, n, os.path.basename(path)))
  print("# measured PMC figures at the same micro-batch (profiles/r03_%s_per_op_counters.txt).  W = weight fragments, read once per launch." % wl)
  print("%-30s %8s %9s %9s | %9s %9s %8s" % ("op", "W MB", "read MB", "write MB", "r3 fetch", "r3 write", "meas/cmp"))
  tot = [0.0] * 5
  for op, rd, wr in rows:
    key = op.split("+")[0]
    wmb = wbytes.get(key, 0) / 1e6
    wmb = min(wmb, rd)
    rdn, wrn = n * rd - (n - 1) * wmb, n * wr
    m = meas.get(op)
    print("%-30s %8.1f %9.1f %9.1f | %9s %9s %8s" % (op[:30], wmb, rdn, wrn, "%.1f" % m[0] if m else "", "%.1f" % m[1] if m else "",
                                                     "%.2f" % ((m[0] + m[1]) / (rdn + wrn)) if m and rdn + wrn > 0 else ""))
    tot[0] += wmb; tot[1] += rdn; tot[2] += wrn
    if m:
      tot[3] += m[0]; tot[4] += m[1]
  cmp_scan, meas_scan = (tot[1] + tot[2]) / n, (tot[3] + tot[4]) / n
  print("%-30s %8.1f %9.1f %9.1f | %9.1f %9.1f %8.2f" % ("TOTAL per %d scans" % n, tot[0], tot[1], tot[2], tot[3], tot[4], (tot[3] + tot[4]) / (tot[1] + tot[2])))
  print("per scan: compulsory %.1f MB, measured %.1f MB" % (cmp_scan, meas_scan))


if __name__ == "__main__":
  main()
