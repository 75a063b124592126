#!/usr/bin/env python3
"""Generate tests/golden/projection_*.npz by RUNNING THE REFERENCE's own spherical projection
(dataset_convert/laserscan_semantic_kitti.py: LaserScan.do_range_projection, pure NumPy) on
seeded synthetic point clouds.  Runs only in the build container (needs /root/reference).

This is the one row whose oracle is pinned by the reference itself: the vectors are its outputs.
Stored per case: points [M,4] float32 (x,y,z,remission) and the reference's proj_range,
proj_xyz, proj_remission, proj_idx (float32 / int32, -1 = no data).
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True   # /root/reference is read-only by contract: no __pycache__ there
REF = "/root/reference/dataset_convert/laserscan_semantic_kitti.py"

CASES = [
  # name, H, W, fov_up, fov_down, points, seed
  ("kitti_64x1024", 64, 1024, 3.0, -25.0, 120000, 11),
  ("small_32x256", 32, 256, 3.0, -25.0, 9000, 12),
  ("nuscenes_like_32x1024", 32, 1024, 10.0, -30.0, 34000, 13),
]


def synth_points(m, fov_up, fov_down, seed):
  """Random returns: mostly inside the vertical field of view, some above/below (clamped rows),
  ranges 1.5 .. 90 m, a few exact duplicates of direction at different ranges (occlusion)."""
  rng = np.random.default_rng(seed)
  yaw = rng.uniform(-np.pi, np.pi, m)
  pitch = np.deg2rad(rng.uniform(fov_down - 3.0, fov_up + 3.0, m))
  r = rng.uniform(1.5, 90.0, m)
  k = m // 10                               # occluded returns: same ray, farther away
  yaw[:k], pitch[:k] = yaw[k:2 * k], pitch[k:2 * k]
  r[:k] = r[k:2 * k] + rng.uniform(0.5, 20.0, k)
  x = r * np.cos(pitch) * np.cos(yaw)
  y = r * np.cos(pitch) * np.sin(yaw)
  z = r * np.sin(pitch)
  rem = rng.uniform(0.0, 1.0, m)
  return np.stack([x, y, z, rem], axis=1).astype(np.float32)


def main():
  spec = importlib.util.spec_from_file_location("ref_laserscan", REF)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  for name, h, w, up, down, m, seed in CASES:
    pts = synth_points(m, up, down, seed)
    scan = mod.LaserScan(project=True, H=h, W=w, fov_up=up, fov_down=down)
    scan.set_points(pts[:, :3].copy(), pts[:, 3].copy())
    np.savez_compressed(os.path.join(HERE, "projection_%s.npz" % name), points=pts,
                        H=h, W=w, fov_up=up, fov_down=down,
                        proj_range=scan.proj_range, proj_xyz=scan.proj_xyz,
                        proj_remission=scan.proj_remission, proj_idx=scan.proj_idx)
    filled = (scan.proj_idx >= 0).mean()
    print(name, "filled %.3f" % filled, "points", m)


if __name__ == "__main__":
  main()
