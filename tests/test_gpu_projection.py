"""Spherical projection row against golden vectors that are OUTPUTS OF
THE REFERENCE's own NumPy code (tests/golden/make_projection_golden.py)."""
import os

import numpy as np
import pytest

from pclsegmentation_amd.projection import LaserScan

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["kitti_64x1024", "small_32x256", "nuscenes_like_32x1024"])
def test_projection_matches_reference(cuda, name):
  g = np.load(os.path.join(GOLDEN, "projection_%s.npz" % name))
  scan = LaserScan(True, int(g["H"]), int(g["W"]), float(g["fov_up"]), float(g["fov_down"]))
  scan.set_points(g["points"][:, :3], g["points"][:, 3])
  idx = scan.proj_idx
  same = idx == g["proj_idx"]
  # NumPy's float32 arctan2/arcsin are not correctly rounded everywhere; a point that sits within
  # an ulp of a cell border may land in the neighbouring cell.  Everything else is exact.
  assert same.mean() >= 0.9995, same.mean()
  assert np.array_equal(scan.proj_range[same], g["proj_range"][same])
  assert np.array_equal(scan.proj_xyz[same], g["proj_xyz"][same])
  assert np.array_equal(scan.proj_remission[same], g["proj_remission"][same])
  # every pixel holds a real point of the cloud (or -1), and its stored depth is that point's norm
  filled = idx >= 0
  pts = g["points"][idx[filled]]
  assert np.array_equal(scan.proj_xyz[filled], pts[:, :3])
  assert np.allclose(scan.proj_range[filled], np.linalg.norm(pts[:, :3], axis=1), rtol=1e-6)


def test_nearest_point_wins_and_empty_value(cuda):
  import torch
  pts = np.array([[10, 0, 0, 0.1], [5, 0, 0, 0.2], [20, 0, 0, 0.3], [0, 0, 0, 0.9]], np.float32)
  scan = LaserScan(True, 16, 64, 3.0, -25.0)
  img = scan.project_device(torch.from_numpy(pts).cuda(), empty=0.0).cpu().numpy()
  idx = scan.proj_idx
  assert (idx >= 0).sum() == 1 and idx.max() == 1          # the 5 m return occludes 10 m and 20 m
  y, x = np.argwhere(idx == 1)[0]
  assert img[y, x].tolist() == [5.0, 0.0, 0.0, np.float32(0.2), 5.0]
  assert (img[idx < 0] == 0.0).all()                        # converter convention for empty pixels
