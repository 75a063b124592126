"""Shared building blocks for the model-config functions.

The reference keeps one hand-written function per config, each returning an
``EasyDict`` (reference: pcl_segmentation/configs/*.py).  ``easydict`` is not a
dependency here: :class:`ModelConfig` is a small attribute-dict with the same
read/write behaviour for the fields the hot path consumes.

Constants (class lists, colour maps, input statistics, grid sizes) are the
reference's values; every table below cites the file:line it restates.
"""
import numpy as np


class ModelConfig(dict):
  """dict with attribute access (``mc.NUM_CLASS`` == ``mc['NUM_CLASS']``)."""

  def __getattr__(self, key):
    try:
      return self[key]
    except KeyError as exc:
      raise AttributeError(key) from exc

  def __setattr__(self, key, value):
    self[key] = value

  def copy(self):
    return ModelConfig(dict.copy(self))


# 11-class ika label set (reference: configs/SqueezeSegV2.py:33-43).
IKA_CLASSES = ["Road", "Sidewalk", "Building", "Pole", "Vegetation", "Person",
               "Two-wheeler", "Car", "Truck", "Bus", "None"]

# The Darknet configs spell one class differently (reference: configs/Darknet53.py:33-43).
IKA_CLASSES_DARKNET = [c if c != "Two-wheeler" else "TwoWheeler" for c in IKA_CLASSES]

# RGB palette of the ika label set (reference: configs/SqueezeSegV2.py:47-58).
IKA_PALETTE_RGB = [
  (128, 64, 128),   # Road
  (244, 35, 232),   # Sidewalk
  (70, 70, 70),     # Building
  (153, 153, 153),  # Pole
  (107, 142, 35),   # Vegetation
  (220, 20, 60),    # Person
  (255, 0, 0),      # Two-wheeler
  (0, 0, 142),      # Car
  (0, 0, 70),       # Truck
  (0, 60, 100),     # Bus
  (0, 0, 0),        # None
]

# 20-class SemanticKITTI label set, "None" FIRST (reference: configs/SqueezeSegV2Kitti.py:35-54).
KITTI_CLASSES = ["None", "car", "bicycle", "motorcycle", "truck", "other-vehicle",
                 "person", "bicyclist", "motorcyclist", "road", "parking", "sidewalk",
                 "other-ground", "building", "fence", "vegetation", "trunk", "terrain",
                 "pole", "traffic-sign"]

# The reference lists this palette as BGR and converts with rgb()
# (reference: configs/SqueezeSegV2Kitti.py:29-30,60-79); kept as BGR here, flipped on use.
KITTI_PALETTE_BGR = [
  (0, 0, 0), (245, 150, 100), (245, 230, 100), (150, 60, 30), (180, 30, 80),
  (255, 0, 0), (30, 30, 255), (200, 40, 255), (90, 30, 150), (255, 0, 255),
  (255, 150, 255), (75, 0, 75), (75, 0, 175), (0, 200, 255), (50, 120, 255),
  (0, 175, 0), (0, 60, 135), (80, 240, 150), (150, 240, 255), (0, 0, 255),
]

# x, y, z, intensity, depth statistics, shape [1,1,5] float64 like the reference.
STATS = {
  # reference: configs/SqueezeSegV2.py:96-97 (also Darknet53.py:91-92, Darknet21.py:91-92)
  "ika": ([24.810, 0.819, 0.000, 16.303, 25.436],
          [30.335, 7.807, 2.058, 25.208, 30.897]),
  # reference: configs/SqueezeSegV2Kitti.py:117-118 (also Darknet53Kitti.py:118-119)
  "kitti": ([-0.047, 0.365, -0.855, 0.2198, 8.3568],
            [10.154, 7.627, 0.8651, 0.1764, 9.6474]),
  # reference: configs/SqueezeSegV2NuScenes.py:98-99
  "nuscenes": ([-0.1090, -0.1645, -0.6275, 17.2574, 11.5727],
               [11.4001, 12.9684, 1.9548, 20.2257, 12.9454]),
}


def palette(colors, dtype, bgr=False):
  """[NC,3] colour map in [0,1]; dtype differs between reference configs
  (float32 for SqueezeSegV2*, float64 for the 11-class Darknet configs)."""
  arr = np.zeros((len(colors), 3), dtype=dtype)
  for i, c in enumerate(colors):
    c = c[::-1] if bgr else c
    arr[i] = np.array(c, dtype) / dtype(255.0)
  return arr


def make_config(*, classes, color_map, h, w, stats, batch_size, loss_weight=None,
                squeezeseg=None, darknet=None, train=None):
  """Assemble one config.  Field names are the reference's."""
  mc = ModelConfig()
  mc.CLASSES = list(classes)
  mc.NUM_CLASS = len(mc.CLASSES)
  mc.CLS_2_ID = dict(zip(mc.CLASSES, range(mc.NUM_CLASS)))
  mc.CLS_LOSS_WEIGHT = (np.ones(mc.NUM_CLASS) if loss_weight is None
                        else np.array(loss_weight, dtype=np.float64))
  mc.CLS_COLOR_MAP = color_map

  # Input shape
  mc.BATCH_SIZE = batch_size
  mc.AZIMUTH_LEVEL = w
  mc.ZENITH_LEVEL = h
  mc.NUM_FEATURES = 6

  # Loss (training-only constants, carried for surface compatibility)
  mc.USE_FOCAL_LOSS = False
  mc.FOCAL_GAMMA = 2.0
  mc.CLS_LOSS_COEF = 15.0
  mc.DENOM_EPSILON = 1e-12

  for group in (train, squeezeseg, darknet):
    if group:
      for k, v in group.items():
        mc[k] = v

  # Dataset augmentation flags (training-only)
  mc.DATA_AUGMENTATION = True
  mc.RANDOM_FLIPPING = True
  mc.SHIFT_UP_DOWN = 0
  mc.SHIFT_LEFT_RIGHT = 70

  mean, std = STATS[stats]
  mc.INPUT_MEAN = np.array([[mean]], dtype=np.float64)
  mc.INPUT_STD = np.array([[std]], dtype=np.float64)
  return mc
