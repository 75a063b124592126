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


# ---------------------------------------------------------------------------------------------
# Second family (tests/golden/projection2_*.npz): the ring-index projection, the label projection
# and the front-view information map — again OUTPUTS OF THE REFERENCE's own NumPy code:
#   dataset_convert/laserscan_nuscenes.py      LaserScan.do_range_projection_ring (:191-223),
#                                              SemLaserScan.do_label_projection (:377-383)
#   dataset_convert/laserscan_semantic_kitti.py SemLaserScan (label projection on the KITTI class)
#   pcl_segmentation/preprocessing/convert_validation_pcd_to_npy.py
#                                              pcl_xyz_i_r_d_l_to_information_map (:97-156)
# Two things the build container lacks are satisfied ONLY so that these modules import; neither is
# touched by the functions exercised here:
#   * `nuscenes` (nuscenes-devkit, file readers PointCloud / load_bin_file), `cv2`, the reference's
#     `configs` package (imports easydict): empty stand-in modules in sys.modules;
#   * `np.float`, an alias NumPy removed in 1.24 and the reference still spells in
#     SemLaserScan.reset(): re-created as the builtin float it always was.
def _load_with_stubs(path, name):
  import types
  for mod, attrs in (("nuscenes", ()), ("nuscenes.utils", ()), ("nuscenes.utils.data_classes", ("PointCloud",)),
                     ("nuscenes.utils.data_io", ("load_bin_file",)), ("cv2", ()), ("configs", ("SqueezeSegV2Config",))):
    if mod not in sys.modules:
      m = types.ModuleType(mod)
      for a in attrs:
        setattr(m, a, type(a, (), {}))
      sys.modules[mod] = m
  if not hasattr(np, "float"):
    np.float = float
  spec = importlib.util.spec_from_file_location(name, path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def synth_ring_sweep(m, rings, seed):
  """nuScenes-like sweep: every point carries the index of the laser ring that produced it; many
  azimuth cells receive several points (later ones overwrite earlier ones in the reference)."""
  rng = np.random.default_rng(seed)
  ring = rng.integers(0, rings, m).astype(np.int32)
  yaw = rng.uniform(-np.pi, np.pi, m)
  pitch = np.deg2rad(-30.0 + 40.0 * ring / (rings - 1)) + rng.normal(0, 0.002, m)
  r = rng.uniform(1.0, 70.0, m)
  pts = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch),
                  rng.uniform(0, 255, m)], axis=1).astype(np.float32)
  labels = rng.integers(0, 32, m).astype(np.uint8)       # lidarseg labels are uint8 class ids
  return pts, ring, labels


def main2():
  nus = _load_with_stubs("/root/reference/dataset_convert/laserscan_nuscenes.py", "ref_laserscan_nuscenes")
  kit = _load_with_stubs(REF, "ref_laserscan_kitti2")
  pcd = _load_with_stubs("/root/reference/pcl_segmentation/preprocessing/convert_validation_pcd_to_npy.py",
                         "ref_convert_pcd")
  color = {i: [i, 2 * i, 3 * i] for i in range(40)}
  # (a) ring projection + label projection, nuScenes geometry 32 x 1024
  for name, h, w, m, seed in (("ring_32x1024", 32, 1024, 34000, 21), ("ring_16x256", 16, 256, 9000, 22)):
    pts, ring, labels = synth_ring_sweep(m, h, seed)
    scan = nus.SemLaserScan(32, color, project=True, H=h, W=w, fov_up=10.0, fov_down=-30.0,
                            use_ring_projection=True)
    scan.set_points(pts[:, :3].copy(), pts[:, 3].copy(), ring.copy())
    scan.set_label(labels.copy())
    np.savez_compressed(os.path.join(HERE, "projection2_%s.npz" % name), points=pts, ring=ring, labels=labels,
                        H=h, W=w, proj_range=scan.proj_range, proj_xyz=scan.proj_xyz,
                        proj_remission=scan.proj_remission, proj_idx=scan.proj_idx,
                        proj_mask=scan.proj_mask, proj_sem_label=scan.proj_sem_label)
    print(name, "filled %.3f" % (scan.proj_idx >= 0).mean())
  # (b) label projection on the elevation-angle (KITTI) class + the converter's learning_map and
  #     final [H,W,6] sample (dataset_convert/semantic_kitti.py:160-171 restated on its outputs)
  pts = synth_points(30000, 3.0, -25.0, 31)
  rng = np.random.default_rng(32)
  labels = rng.choice(np.array([0, 1, 10, 11, 13, 15, 18, 20, 30, 31, 32, 40, 44, 48, 49, 50, 51, 52, 60, 70, 71,
                                72, 80, 81, 99, 252, 253, 254, 255, 256, 257, 258, 259], np.uint32), 30000)
  learning_map = {int(k): int(i % 20) for i, k in enumerate(np.unique(labels))}
  learning_map[0] = 0
  scan = kit.SemLaserScan(20, {int(k): [1, 2, 3] for k in learning_map}, project=True, H=32, W=512)
  scan.set_points(pts[:, :3].copy(), pts[:, 3].copy())
  scan.set_label(labels.copy())
  mask = scan.proj_range > 0
  rng_img, xyz, rem = scan.proj_range.copy(), scan.proj_xyz.copy(), scan.proj_remission.copy()
  rng_img[~mask] = 0.0
  xyz[~mask] = 0.0
  rem[~mask] = 0.0
  mapped = np.vectorize(learning_map.get)(scan.proj_sem_label)
  final = np.concatenate([xyz, rem.reshape(32, 512, 1), rng_img.reshape(32, 512, 1), mapped.reshape(32, 512, 1)], axis=2)
  keys = np.array(sorted(learning_map), np.int64)
  np.savez_compressed(os.path.join(HERE, "projection2_kitti_labels_32x512.npz"), points=pts, labels=labels,
                      H=32, W=512, fov_up=3.0, fov_down=-25.0, proj_idx=scan.proj_idx,
                      proj_sem_label=scan.proj_sem_label, map_keys=keys,
                      map_values=np.array([learning_map[int(k)] for k in keys], np.int64), final=final)
  # (c) front-view information map of the validation converter
  rng = np.random.default_rng(41)
  m = 12000
  phi = rng.uniform(-0.55, 0.6, m)                 # beyond the +-24 deg window on both sides
  r = rng.uniform(1.0, 60.0, m)
  ring = rng.integers(0, 32, m)
  pitch = np.deg2rad(-15.0 + ring)
  x = (r * np.cos(pitch) * np.cos(phi)).astype(np.float32)
  y = (r * np.cos(pitch) * np.sin(phi)).astype(np.float32)
  z = (r * np.sin(pitch)).astype(np.float32)
  d = np.sqrt(x.astype(np.float64) ** 2 + y ** 2 + z ** 2).astype(np.float32)
  d[rng.random(m) < 0.05] = 0.0                    # invalid returns: mask 0
  pcl = np.stack([x, y, z, rng.uniform(0, 1, m).astype(np.float32), ring.astype(np.float32), d,
                  rng.integers(0, 12, m).astype(np.float32)], axis=1).astype(np.float64)
  info = pcd.pcl_xyz_i_r_d_l_to_information_map(pcl.copy(), H=32, W=240, C=7)
  np.savez_compressed(os.path.join(HERE, "projection2_front_32x240.npz"), pcl=pcl, info=info)
  print("front map filled %.3f" % (info[..., 6] > 0).mean())


if __name__ == "__main__":
  main2()
