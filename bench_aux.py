#!/usr/bin/env python3
"""Measurement of the two rows built around the forward pass to
the same bar as bench.py: inputs resident in HBM, HIP events on the launch stream, algorithmic
bytes against the 8 TB/s HBM peak, and the CPU restatement (oracle, NumPy) timed beside it on a
bounded sample.  One JSON line per row.  Not part of the driver's bench contract (bench.py is).

  projection        KITTI-size sweep [M = 120k points, 4 floats] -> range image [64, 2048, 5]
                    ALG bytes = 16 B/point read + 20 B/pixel written (+ 4 B/pixel index map)
  confusion_matrix  labels + predictions of 32 scans 64x2048, NUM_CLASS 20 -> int64 [20, 20]
                    ALG bytes = 8 B/pixel read
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def timed(fn, stream, steps, warmup):
  import torch
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(stream)
  for _ in range(steps):
    fn()
  e1.record(stream)
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e-3 / steps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--cpu-seconds", type=float, default=5.0)
  args = ap.parse_args()
  import torch
  from pclsegmentation_amd import engine as E
  from oracle import np_oracle as O      # CPU baseline leg only
  if not torch.cuda.is_available():
    raise SystemExit("bench_aux.py needs an MI355X; there is no CPU fallback")
  dev = torch.device("cuda", 0)
  stream = torch.cuda.current_stream(dev)
  rng = np.random.default_rng(1234)

  # ---- spherical projection: a synthetic 64-beam sweep, KITTI field of view (3, -25) degrees
  H, W, M = 64, 2048, 120000
  yaw = rng.uniform(-np.pi, np.pi, M)
  pitch = np.deg2rad(rng.uniform(-24.5, 2.5, M))
  r = rng.uniform(2.0, 80.0, M)
  pts = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch),
                  rng.uniform(0, 1, M)], -1).astype(np.float32)
  d_pts = torch.from_numpy(pts).to(dev)
  d_img = torch.empty((H, W, 5), dtype=torch.float32, device=dev)
  d_idx = torch.empty((H, W), dtype=torch.int32, device=dev)
  d_scr = torch.empty((H * W,), dtype=torch.int64, device=dev)
  t = timed(lambda: E.op_project(d_pts, M, H, W, 3.0, -25.0, -1.0, d_img, d_idx, d_scr, stream.cuda_stream),
            stream, args.steps, args.warmup)
  alg = M * 16 + H * W * 24
  n = 0
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < args.cpu_seconds:
    O.range_projection(pts, H, W, 3.0, -25.0)
    n += 1
  cpu = n / (time.perf_counter() - t0)
  print(json.dumps({
    "metric": "LiDAR sweeps/sec spherical projection (120k points -> 64x2048x5)", "value": round(1.0 / t, 1),
    "unit": "sweeps/s", "us_per_sweep": round(t * 1e6, 2), "dtype": "f32 (f64 angles)", "data": "synthetic",
    "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes": alg,
                 "note": "3 launches (init, atomicMin scatter, gather); 4.5 MB per sweep: launch-latency bound, not bandwidth"},
    "cpu_baseline": {"value": round(cpu, 2), "unit": "sweeps/s", "cores": 1, "kind": "port",
                     "sample": "%d sweeps, NumPy restatement of LaserScan.do_range_projection (oracle/np_oracle.py)" % n}}),
    flush=True)

  # ---- confusion matrix
  N, NC = 32, 20
  P = N * H * W
  labels = rng.integers(0, NC, P, dtype=np.int32)
  preds = np.where(rng.random(P) < 0.8, labels, rng.integers(0, NC, P, dtype=np.int32)).astype(np.int32)
  d_l, d_p = torch.from_numpy(labels).to(dev), torch.from_numpy(preds).to(dev)
  d_cm = torch.zeros((NC, NC), dtype=torch.int64, device=dev)
  t = timed(lambda: E.op_confusion_matrix(d_l, d_p, P, NC, d_cm, stream.cuda_stream), stream, args.steps, args.warmup)
  alg = P * 8
  n = 0
  t0 = time.perf_counter()
  sample = 4 * H * W
  while time.perf_counter() - t0 < args.cpu_seconds:
    O.confusion_matrix(labels[:sample], preds[:sample], NC)
    n += 1
  cpu = n * 4 / (time.perf_counter() - t0)
  print(json.dumps({
    "metric": "LiDAR scans/sec confusion-matrix accumulation (64x2048, NC 20)", "value": round(N / t, 1),
    "unit": "scans/s", "us_per_batch": round(t * 1e6, 2), "dtype": "int32 -> int64 counts", "data": "synthetic",
    "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes": alg},
    "cpu_baseline": {"value": round(cpu, 1), "unit": "scans/s", "cores": 1, "kind": "port",
                     "sample": "%d x 4 scans, NumPy bincount restatement (oracle/np_oracle.py)" % n}}), flush=True)


if __name__ == "__main__":
  main()
