"""Spherical projection on the device: LiDAR sweep -> range image, the step before the network.
Host-side mirror of the reference's converters' scan classes, keeping their names and attributes:

  LaserScan / SemLaserScan   dataset_convert/laserscan_semantic_kitti.py:106-166 (elevation rows, the
                             nearest point wins) and dataset_convert/laserscan_nuscenes.py:191-288
                             (``use_ring_projection``: row = H-1-ring_index, the LAST point wins),
                             label projection :377-383, used by semantic_kitti.py:150-179 and
                             nu_dataset.py:128-173 to write the [H,W,6] .npy samples the network reads
  pcl_xyz_i_r_d_l_to_information_map
                             preprocessing/convert_validation_pcd_to_npy.py:97-156 (front-view window,
                             ring rows, channels x,y,z,i,d,label,mask)

All arithmetic runs in libpclseg.so (pclseg_op_project_ex); with this step on the GPU a raw sweep
goes straight into ``model.predict_raw`` without the offline .npy stage."""
import numpy as np

from . import engine as _engine


class LaserScan:
  """Counterpart of the reference's ``LaserScan`` for the projection: same constructor arguments
  and the same result attributes (``proj_range``, ``proj_xyz``, ``proj_remission``, ``proj_idx``,
  ``proj_mask``; -1 = no data)."""

  def __init__(self, project=True, H=64, W=1024, fov_up=3.0, fov_down=-25.0, use_ring_projection=False,
               device=0):
    import torch
    self.project = project
    self.proj_H, self.proj_W = int(H), int(W)
    self.proj_fov_up, self.proj_fov_down = fov_up, fov_down
    self.use_ring_projection = bool(use_ring_projection)
    self._dev = _engine.torch_device(device)
    self._scratch = torch.empty(self.proj_H * self.proj_W, dtype=torch.int64, device=self._dev)
    self.image = None
    self._idx = None
    self.points = np.zeros((0, 3), np.float32)
    self.remissions = np.zeros((0,), np.float32)
    self.ring_index = np.zeros((0,), np.int32)

  def size(self):
    return self.points.shape[0]

  def __len__(self):
    return self.size()

  def set_points(self, points, remissions=None, ring_index=None):
    """reference: laserscan_nuscenes.py:150-189 — ring projection when a ring index is given and
    ``use_ring_projection`` is set, elevation projection when fov_up/fov_down are set, else
    NotImplementedError."""
    import torch
    if not isinstance(points, np.ndarray):
      raise TypeError("Scan should be numpy array")
    pts = np.asarray(points, np.float32)
    self.points = pts[:, :3]
    self.remissions = (np.zeros(pts.shape[0], np.float32) if remissions is None
                       else np.asarray(remissions, np.float32))
    self.ring_index = (np.zeros(pts.shape[0], np.int32) if ring_index is None
                       else np.asarray(ring_index, np.int32))
    if not self.project:
      return
    packed = np.ascontiguousarray(np.concatenate([self.points, self.remissions.reshape(-1, 1)], axis=1))
    d_pts = torch.from_numpy(packed).to(self._dev)
    if ring_index is not None and self.use_ring_projection:
      self.project_device(d_pts, empty=-1.0, ring_dev=torch.from_numpy(self.ring_index).to(self._dev))
    elif self.proj_fov_up is not None and self.proj_fov_down is not None and not self.use_ring_projection:
      self.project_device(d_pts, empty=-1.0)
    else:
      raise NotImplementedError

  def project_device(self, points_dev, empty=0.0, ring_dev=None, labels_dev=None, lut_dev=None,
                     out_channels=5):
    """points_dev: torch float32 [M,>=4] on the device -> torch [H,W,out_channels]
    (x,y,z,remission,depth[,label]).  ``ring_dev`` (int32 [M]) selects the ring projection."""
    import torch
    h, w = self.proj_H, self.proj_W
    points_dev = points_dev.contiguous()
    self.image = torch.empty((h, w, out_channels), dtype=torch.float32, device=self._dev)
    self._idx = torch.empty((h, w), dtype=torch.int32, device=self._dev)
    ring = ring_dev is not None
    desc = _engine.make_proj_desc(h, w, row_mode=_engine.PROJ_ROW_RING if ring else _engine.PROJ_ROW_FOV,
                                  winner=_engine.PROJ_LAST if ring else _engine.PROJ_NEAREST,
                                  out_channels=out_channels, fov_up=self.proj_fov_up or 0.0,
                                  fov_down=self.proj_fov_down or 0.0, empty=empty)
    _engine.op_project_ex(desc, points_dev, points_dev.shape[1], points_dev.shape[0], ring_dev, None,
                          labels_dev, lut_dev, self.image, self._idx, self._scratch,
                          _engine.stream_handle(self._dev))
    self._points_dev = points_dev
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

  @property
  def proj_mask(self):
    """reference quirk kept: ``(proj_idx > 0)`` — the pixel won by point 0 counts as empty
    (laserscan_nuscenes.py:223,288)."""
    return (self.proj_idx > 0).astype(np.float32)


class SemLaserScan(LaserScan):
  """``LaserScan`` + semantic labels (reference: laserscan_nuscenes.py:291-383,
  laserscan_semantic_kitti.py SemLaserScan).  ``proj_sem_label`` is gathered on the device through
  ``proj_idx``; pixels without a point keep label 0."""

  def __init__(self, nclasses=None, sem_color_dict=None, project=False, H=64, W=1024, fov_up=3.0,
               fov_down=-25.0, use_ring_projection=False, device=0):
    super(SemLaserScan, self).__init__(project, H, W, fov_up, fov_down, use_ring_projection, device)
    self.nclasses = nclasses
    self.sem_label = np.zeros((0,), np.int32)
    self._label_image = None
    if sem_color_dict:
      max_key = max(sem_color_dict) + 1
      self.sem_color_lut = np.zeros((max_key + 100, 3), np.float32)
      for key, value in sem_color_dict.items():
        self.sem_color_lut[key] = np.array(value, np.float32) / 255.0
    else:
      self.sem_color_lut = None

  def set_label(self, label, learning_map=None):
    """reference: set_label + do_label_projection.  ``learning_map`` (dict or int array) is the
    converters' label -> train-id table (semantic_kitti.py:145,165), applied to the projected image
    on the device; without it the raw labels are projected."""
    import torch
    if not isinstance(label, np.ndarray):
      raise TypeError("Label should be numpy array")
    if label.shape[0] != self.points.shape[0]:
      raise ValueError("Scan and Label don't contain same number of points")
    self.sem_label = np.asarray(label).astype(np.int32)
    if not self.project:
      return
    lut = None
    if learning_map is not None:
      if isinstance(learning_map, dict):
        arr = np.full(max(learning_map) + 1, -1, np.int32)
        for k, v in learning_map.items():
          arr[k] = v
        learning_map = arr
      lut = torch.from_numpy(np.ascontiguousarray(learning_map, np.int32)).to(self._dev)
    ring = torch.from_numpy(self.ring_index).to(self._dev) if self.use_ring_projection else None
    img = self.project_device(self._points_dev, empty=-1.0, ring_dev=ring,
                              labels_dev=torch.from_numpy(self.sem_label).to(self._dev), lut_dev=lut,
                              out_channels=6)
    self._label_image = img[..., 5]

  @property
  def proj_sem_label(self):
    return self._label_image.cpu().numpy().astype(np.int32)

  @property
  def proj_sem_color(self):
    return self.sem_color_lut[self.proj_sem_label] * (self.proj_idx >= 0)[..., None]

  def sample(self):
    """The converters' final [H,W,6] array: xyz, remission, range (0 where empty) and the label
    channel (semantic_kitti.py:160-171, nu_dataset.py:157-167)."""
    out = self.image.clone()
    out[..., :5] = out[..., :5] * (out[..., 4:5] > 0)
    return out.cpu().numpy()


def pcl_xyz_i_r_d_l_to_information_map(pcl, H=32, W=240, C=7, leftPhi=np.radians(24.32),
                                        rightPhi=np.radians(22.23), device=0):
  """reference: preprocessing/convert_validation_pcd_to_npy.py:97-156.  ``pcl`` [M,7] = x, y, z,
  intensity, ring, depth, label -> information map [H,W,7] = x, y, z, i, d, label, mask (float64
  like the reference's np.zeros), front-view azimuth window, ring rows, last point wins."""
  import torch
  if C != 7:
    raise ValueError("the reference writes 7 channels")
  pcl = np.asarray(pcl)
  dev = _engine.torch_device(device)
  pts = torch.from_numpy(np.ascontiguousarray(pcl[:, :4], np.float32)).to(dev)
  ring = torch.from_numpy(np.ascontiguousarray(pcl[:, 4].astype(int), np.int32)).to(dev)
  depth = torch.from_numpy(np.ascontiguousarray(pcl[:, 5], np.float32)).to(dev)
  label = torch.from_numpy(np.ascontiguousarray(pcl[:, 6].astype(int), np.int32)).to(dev)
  image = torch.empty((H, W, 7), dtype=torch.float32, device=dev)
  scratch = torch.empty(H * W, dtype=torch.int64, device=dev)
  desc = _engine.make_proj_desc(H, W, row_mode=_engine.PROJ_ROW_RING, col_mode=_engine.PROJ_COL_FRONT,
                                winner=_engine.PROJ_LAST, out_channels=7, left_phi=float(leftPhi),
                                right_phi=float(rightPhi), empty=0.0)
  _engine.op_project_ex(desc, pts, 4, pts.shape[0], ring, depth, label, None, image, None, scratch,
                        _engine.stream_handle(dev))
  return image.cpu().numpy().astype(np.float64)
