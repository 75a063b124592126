"""Model configs — same function names and fields as the reference's
``pcl_segmentation/configs`` package (one function per config, returning an
attribute-dict of constants).

reference: configs/SqueezeSegV2.py:30, configs/SqueezeSegV2Kitti.py:32,
configs/SqueezeSegV2NuScenes.py:30, configs/Darknet53.py:30, configs/Darknet21.py:30,
configs/Darknet53Kitti.py:32.  The reference's ``configs/__init__.py:1`` exports only
``SqueezeSegV2Config``; all six are exported here.
"""
import numpy as np

from ._base import (ModelConfig, make_config, palette, IKA_CLASSES, IKA_CLASSES_DARKNET,
                    IKA_PALETTE_RGB, KITTI_CLASSES, KITTI_PALETTE_BGR)

__all__ = ["ModelConfig", "SqueezeSegV2Config", "SqueezeSegV2KittiConfig",
           "SqueezeSegV2ConfigNuScenes", "Darknet53", "Darknet21", "Darknet53Kitti"]


def _ssv2_net(bn_momentum):
  return dict(L2_WEIGHT_DECAY=0.05, DROP_RATE=0.1, BN_MOMENTUM=bn_momentum, REDUCTION=16)


def _darknet_net(num_layers):
  # OUTPUT_STRIDE is horizontal only (reference: configs/Darknet53.py:81-82)
  return dict(DROP_RATE=0.01, BN_MOMENTUM=0.9, NUM_LAYERS=num_layers, OUTPUT_STRIDE=16)


def _train(lr, steps, factor, clip):
  return dict(LEARNING_RATE=lr, LR_DECAY_STEPS=steps, LR_DECAY_FACTOR=factor, MAX_GRAD_NORM=clip)


def SqueezeSegV2Config():
  """ika 11-class, 32x240 (reference: configs/SqueezeSegV2.py:30-99)."""
  return make_config(classes=IKA_CLASSES, color_map=palette(IKA_PALETTE_RGB, np.float32),
                     h=32, w=240, stats="ika", batch_size=32,
                     squeezeseg=_ssv2_net(0.99), train=_train(0.003, 1000, 0.97, 100.0))


def SqueezeSegV2KittiConfig():
  """SemanticKITTI 20-class, 64x1024 (reference: configs/SqueezeSegV2Kitti.py:32-120)."""
  return make_config(classes=KITTI_CLASSES,
                     color_map=palette(KITTI_PALETTE_BGR, np.float32, bgr=True),
                     h=64, w=1024, stats="kitti", batch_size=64,
                     squeezeseg=_ssv2_net(0.9), train=_train(0.001, 500, 0.99, 100.0))


def SqueezeSegV2ConfigNuScenes():
  """nuScenes 11-class, 32x1024; the "None" class has zero loss weight
  (reference: configs/SqueezeSegV2NuScenes.py:30-101)."""
  return make_config(classes=IKA_CLASSES, color_map=palette(IKA_PALETTE_RGB, np.float32),
                     h=32, w=1024, stats="nuscenes", batch_size=32,
                     loss_weight=[1.0] * 10 + [0.0],
                     squeezeseg=_ssv2_net(0.99), train=_train(0.003, 1000, 0.99, 100.0))


def Darknet53():
  """ika 11-class, 32x240, 53 layers (reference: configs/Darknet53.py:30-94)."""
  return make_config(classes=IKA_CLASSES_DARKNET, color_map=palette(IKA_PALETTE_RGB, np.float64),
                     h=32, w=240, stats="ika", batch_size=16,
                     darknet=_darknet_net(53), train=_train(0.005, 500, 0.99, 1.0))


def Darknet21():
  """ika 11-class, 32x240, 21 layers (reference: configs/Darknet21.py:30-94)."""
  return make_config(classes=IKA_CLASSES_DARKNET, color_map=palette(IKA_PALETTE_RGB, np.float64),
                     h=32, w=240, stats="ika", batch_size=16,
                     darknet=_darknet_net(21), train=_train(0.01, 500, 0.99, 1.0))


def Darknet53Kitti():
  """SemanticKITTI 20-class, 64x1024, 53 layers (reference: configs/Darknet53Kitti.py:32-121)."""
  return make_config(classes=KITTI_CLASSES,
                     color_map=palette(KITTI_PALETTE_BGR, np.float32, bgr=True),
                     h=64, w=1024, stats="kitti", batch_size=16,
                     darknet=_darknet_net(53), train=_train(0.001, 500, 0.99, 100.0))
