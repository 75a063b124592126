"""Spherical projection row against golden vectors that are OUTPUTS OF
THE REFERENCE's own NumPy code (tests/golden/make_projection_golden.py)."""
import os

import numpy as np
import pytest

from pclsegmentation_amd.projection import LaserScan

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _border_points(points, H, W, fov_up, fov_down):
  """Points whose continuous image coordinates (float64) sit within a few float32 ulps of a cell
  border: only for these can NumPy's float32 arctan2 / arcsin (not correctly rounded, platform
  dependent) and the kernel's float64-then-round-once evaluation pick different cells."""
  p = points.astype(np.float64)
  depth = np.sqrt((p[:, :3] ** 2).sum(1))
  up, down = fov_up / 180.0 * np.pi, fov_down / 180.0 * np.pi
  cx = 0.5 * (-np.arctan2(p[:, 1], p[:, 0]) / np.pi + 1.0) * W
  cy = (1.0 - (np.arcsin(p[:, 2] / depth) + abs(down)) / (abs(down) + abs(up))) * H
  tol_x = 8 * np.spacing(np.float32(W))      # ~4 float32 operations on values up to W
  tol_y = 8 * np.spacing(np.float32(H))
  near = lambda c, tol: np.abs(c - np.round(c)) < tol
  return near(cx, tol_x) | near(cy, tol_y)


@pytest.mark.parametrize("name", ["kitti_64x1024", "small_32x256", "nuscenes_like_32x1024"])
def test_projection_matches_reference(cuda, name):
  g = np.load(os.path.join(GOLDEN, "projection_%s.npz" % name))
  H, W = int(g["H"]), int(g["W"])
  scan = LaserScan(True, H, W, float(g["fov_up"]), float(g["fov_down"]))
  scan.set_points(g["points"][:, :3], g["points"][:, 3])
  idx, gold = scan.proj_idx, g["proj_idx"]
  same = idx == gold
  # Index work is exact EXCEPT where a point sits on a cell border (see _border_points): the
  # mismatching pixels are enumerated, each must involve such a border point, and there are few.
  border = _border_points(g["points"], H, W, float(g["fov_up"]), float(g["fov_down"]))
  bad = np.argwhere(~same)
  print("%s: %d of %d pixels differ from the reference (%d border points in %d)"
        % (name, len(bad), same.size, int(border.sum()), len(border)))
  for y, x in bad:
    winners = [i for i in (idx[y, x], gold[y, x]) if i >= 0]
    assert any(border[i] for i in winners), ("non-border mismatch at", y, x, idx[y, x], gold[y, x])
  assert len(bad) <= 2 * border.sum() and len(bad) <= 5e-4 * same.size
  assert np.array_equal(scan.proj_range[same], g["proj_range"][same])
  assert np.array_equal(scan.proj_xyz[same], g["proj_xyz"][same])
  assert np.array_equal(scan.proj_remission[same], g["proj_remission"][same])
  # every pixel holds a real point of the cloud (or -1), and its stored depth is that point's norm
  filled = idx >= 0
  pts = g["points"][idx[filled]]
  assert np.array_equal(scan.proj_xyz[filled], pts[:, :3])
  assert np.allclose(scan.proj_range[filled], np.linalg.norm(pts[:, :3], axis=1), rtol=1e-6)


@pytest.mark.parametrize("name", ["ring_32x1024", "ring_16x256"])
def test_ring_projection_and_labels_match_reference(cuda, name):
  """laserscan_nuscenes.py:191-223 (row = H-1-ring, the LAST point of a pixel wins) and the label
  projection :377-383, against outputs of the reference's own code.  Rows are exact by
  construction; columns can differ only for azimuth-border points (enumerated)."""
  from pclsegmentation_amd.projection import SemLaserScan
  g = np.load(os.path.join(GOLDEN, "projection2_%s.npz" % name))
  H, W = int(g["H"]), int(g["W"])
  scan = SemLaserScan(32, None, project=True, H=H, W=W, fov_up=10.0, fov_down=-30.0, use_ring_projection=True)
  scan.set_points(g["points"][:, :3], g["points"][:, 3], g["ring"])
  scan.set_label(g["labels"])
  idx, gold = scan.proj_idx, g["proj_idx"]
  same = idx == gold
  p = g["points"].astype(np.float64)
  cx = 0.5 * (-np.arctan2(p[:, 1], p[:, 0]) / np.pi + 1.0) * W
  border = np.abs(cx - np.round(cx)) < 8 * np.spacing(np.float32(W))
  bad = np.argwhere(~same)
  print("%s: %d of %d pixels differ (%d border points)" % (name, len(bad), same.size, int(border.sum())))
  for y, x in bad:
    assert any(border[i] for i in (idx[y, x], gold[y, x]) if i >= 0)
  assert len(bad) <= 2 * border.sum() + 1
  assert np.array_equal(scan.proj_range[same], g["proj_range"][same])
  assert np.array_equal(scan.proj_xyz[same], g["proj_xyz"][same])
  assert np.array_equal(scan.proj_remission[same], g["proj_remission"][same])
  assert np.array_equal(scan.proj_sem_label[same], g["proj_sem_label"][same])
  assert np.array_equal(scan.proj_mask[same], g["proj_mask"][same])
  # size-independent property: the winner of a pixel is the LAST point (highest index) of that
  # pixel among the points the kernel itself placed there
  filled = idx >= 0
  assert np.array_equal(scan.proj_xyz[filled], g["points"][idx[filled], :3])
  rows = (H - 1) - g["ring"]
  assert np.array_equal(np.argwhere(filled)[:, 0], rows[idx[filled]])


def test_label_map_and_converter_sample_match_reference(cuda):
  """The converters' final [H,W,6] sample (semantic_kitti.py:160-171): xyz / remission / range with
  empty pixels zeroed and the label channel passed through learning_map."""
  from pclsegmentation_amd.projection import SemLaserScan
  g = np.load(os.path.join(GOLDEN, "projection2_kitti_labels_32x512.npz"))
  lm = {int(k): int(v) for k, v in zip(g["map_keys"], g["map_values"])}
  scan = SemLaserScan(20, None, project=True, H=int(g["H"]), W=int(g["W"]), fov_up=float(g["fov_up"]),
                      fov_down=float(g["fov_down"]))
  scan.set_points(g["points"][:, :3], g["points"][:, 3])
  scan.set_label(g["labels"].astype(np.int64), learning_map=lm)
  idx, gold = scan.proj_idx, g["proj_idx"]
  same = idx == gold
  # enumerate the differing pixels: each must involve a point on a float32 cell border (_border_points)
  border = _border_points(g["points"], int(g["H"]), int(g["W"]), float(g["fov_up"]), float(g["fov_down"]))
  bad = np.argwhere(~same)
  print("kitti labels: %d of %d pixels differ from the reference (%d border points)" % (len(bad), same.size, int(border.sum())))
  for y, x in bad:
    assert any(border[i] for i in (idx[y, x], gold[y, x]) if i >= 0), ("non-border mismatch at", y, x)
  assert len(bad) <= 2 * border.sum()
  final = scan.sample()
  assert final.shape == g["final"].shape
  assert np.array_equal(final[same].astype(np.float64), g["final"][same])


def test_front_view_information_map_matches_reference(cuda):
  """preprocessing/convert_validation_pcd_to_npy.py:97-156: front window, truncating column index,
  out-of-window points dropped, ring rows, last point wins, 7 channels incl. mask = depth > 0."""
  from pclsegmentation_amd.projection import pcl_xyz_i_r_d_l_to_information_map
  f = np.load(os.path.join(GOLDEN, "projection2_front_32x240.npz"))
  got = pcl_xyz_i_r_d_l_to_information_map(f["pcl"], H=32, W=240, C=7)
  want = f["info"]
  assert got.shape == want.shape and got.dtype == np.float64
  diff = np.argwhere((got != want).any(-1))
  # float64 atan2 on both sides; a pixel may differ only where a point's continuous column
  # (left_phi - atan2(y, x)) / dphi sits within a few float64 ulps of an integer (device libm vs glibc):
  # enumerate the differing pixels and find such a point in the pixel or its column neighbours
  print("front map: %d of %d pixels differ" % (len(diff), 32 * 240))
  pcl = f["pcl"].astype(np.float64)
  left, right = np.radians(24.32), np.radians(22.23)       # convert_validation_pcd_to_npy.py:99-100
  col = (left - np.arctan2(pcl[:, 1], pcl[:, 0])) / ((right + left) / 240)
  on_border = np.abs(col - np.round(col)) < 64 * np.spacing(240.0)
  for y, x in diff:
    cands = on_border & (np.abs(np.round(col) - x) <= 1)
    assert cands.any(), ("front-view mismatch without a column-border point", y, x)
  assert len(diff) <= 2 * max(1, int(on_border.sum()))
  ok = ~(got != want).any(-1)
  assert np.array_equal(got[ok], want[ok])
  assert np.array_equal(got[..., 6], (got[..., 4] > 0).astype(np.float64))


def test_nearest_point_wins_and_empty_value(cuda):
  import torch
  pts = np.array([[10, 0, 0, 0.1], [5, 0, 0, 0.2], [20, 0, 0, 0.3], [0, 0, 0, 0.9]], np.float32)
  scan = LaserScan(True, 16, 64, 3.0, -25.0)
  img = scan.project_device(torch.from_numpy(pts).to(cuda), empty=0.0).cpu().numpy()
  idx = scan.proj_idx
  assert (idx >= 0).sum() == 1 and idx.max() == 1          # the 5 m return occludes 10 m and 20 m
  y, x = np.argwhere(idx == 1)[0]
  assert img[y, x].tolist() == [5.0, 0.0, 0.0, np.float32(0.2), 5.0]
  assert (img[idx < 0] == 0.0).all()                        # converter convention for empty pixels
