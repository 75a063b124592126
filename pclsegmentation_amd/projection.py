"""Spherical projection on the device: LiDAR sweep [M,4] -> range image [H,W,5]
(reference: dataset_convert/laserscan_semantic_kitti.py:106-166, used by the converters
dataset_convert/semantic_kitti.py:150-179 to write the .npy scans the network consumes).
With this step on the GPU a raw sweep goes straight into ``model.predict_raw`` without the
offline .npy stage."""
import numpy as np

from . import engine as _engine


class LaserScan:
  """Minimal counterpart of the reference's ``LaserScan`` for the projection only: same
  constructor arguments and the same result attributes (``proj_range``, ``proj_xyz``,
  ``proj_remission``, ``proj_idx``; -1 = no data)."""

  def __init__(self, project=True, H=64, W=1024, fov_up=3.0, fov_down=-25.0, device=0):
    import torch
    self.project = project
    self.proj_H, self.proj_W = int(H), int(W)
    self.proj_fov_up, self.proj_fov_down = float(fov_up), float(fov_down)
    self._dev = torch.device("cuda", device)
    self._scratch = torch.empty(self.proj_H * self.proj_W, dtype=torch.int64, device=self._dev)
    self.image = None
    self._idx = None

  def set_points(self, points, remissions=None):
    import torch
    pts = np.asarray(points, np.float32)
    if remissions is None:
      remissions = np.zeros(pts.shape[0], np.float32)
    packed = np.concatenate([pts[:, :3], np.asarray(remissions, np.float32).reshape(-1, 1)], axis=1)
    self.project_device(torch.from_numpy(np.ascontiguousarray(packed)).to(self._dev), empty=-1.0)

  def project_device(self, points_dev, empty=0.0):
    """points_dev: torch float32 [M,4] on the device -> torch [H,W,5] (x,y,z,remission,depth)."""
    import torch
    h, w = self.proj_H, self.proj_W
    self.image = torch.empty((h, w, 5), dtype=torch.float32, device=self._dev)
    self._idx = torch.empty((h, w), dtype=torch.int32, device=self._dev)
    _engine.op_project(points_dev.contiguous(), points_dev.shape[0], h, w, self.proj_fov_up,
                       self.proj_fov_down, empty, self.image, self._idx, self._scratch,
                       torch.cuda.current_stream(self._dev).cuda_stream)
    return self.image

  @property
  def proj_xyz(self):
    return self.image[..., :3].cpu().numpy()

  @property
  def proj_remission(self):
    return self.image[..., 3].cpu().numpy()

  @property
  def proj_range(self):
    return self.image[..., 4].cpu().numpy()

  @property
  def proj_idx(self):
    return self._idx.cpu().numpy()
