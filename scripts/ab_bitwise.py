#!/usr/bin/env python3
"""Do two builds of the library produce BIT-IDENTICAL outputs?  Kernel variants that only re-order
memory traffic, not arithmetic: every accumulator still receives the same products in the same order, so logits and
predictions must equal the shipped library's bit for bit on all three networks.

usage: ab_bitwise.py <libA.so> <libB.so>            (runs itself once per library in a child process)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("squeezesegv2", "squeezesegv2kitti", 64, 2048, 6, 0.78), ("darknet21", "darknet21", 32, 1024, 5, 0.59),
         ("darknet53", "darknet53kitti", 64, 2048, 3, 0.78), ("squeezesegv2", "squeezesegv2", 32, 240, 7, 0.84)]


def child(out):
  sys.path.insert(0, ROOT)
  import pclsegmentation_amd as P
  from pclsegmentation_amd import engine as E
  from pclsegmentation_amd.utils.synthetic import synthetic_scans
  res = {}
  for model_name, cfg, h, w, n, pv in CASES:
    mc, model = P.load_model_config(model_name, cfg, height=h, width=w)
    model.init_weights(4321)
    raw = synthetic_scans(n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, pv, seed=5)
    preds = np.empty((n, h, w), np.int32)
    logits = np.empty((n, h, w, mc.NUM_CLASS), np.float32)
    model.engine(h, w).forward_raw(raw, n, preds, None, logits, None, mem=E.MEM_HOST)
    res["%s_%dx%d_preds" % (model_name, h, w)] = preds
    res["%s_%dx%d_logits" % (model_name, h, w)] = logits
    model._drop_engines()
  np.savez(out, **res)


def main():
  if sys.argv[1] == "--child":
    return child(sys.argv[2])
  libs = [os.path.abspath(p) for p in sys.argv[1:3]]
  outs = []
  with tempfile.TemporaryDirectory() as tmp:
    for i, lib in enumerate(libs):
      out = os.path.join(tmp, "out%d.npz" % i)
      env = dict(os.environ, PCLSEG_LIB=lib, PCLSEG_DEBUG="1")
      subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out], env=env)
      outs.append(dict(np.load(out)))
  bad = 0
  for k in sorted(outs[0]):
    a, b = outs[0][k], outs[1][k]
    same = a.shape == b.shape and np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    d = 0.0 if same or a.dtype != np.float32 else float(np.abs(a.astype(np.float64) - b).max())
    print("%-36s %s%s" % (k, "bit-identical" if same else "DIFFERS", "" if same else " (max abs diff %.3g, %d elements)" % (d, int((a != b).sum()))))
    bad += not same
  print("A = %s\nB = %s\n%s" % (libs[0], libs[1], "ALL BIT-IDENTICAL" if not bad else "%d arrays differ" % bad))
  return 1 if bad else 0


if __name__ == "__main__":
  sys.exit(main())
