"""Name -> class registries and ``load_model_config`` (reference: utils/args_loader.py:36-55).

Same keys, same behaviour: lookups are ``dict[name.lower()]`` so an unknown name raises
``KeyError``.  ``load_model_config`` additionally accepts ``height=`` / ``width=``
overrides because the benchmark shapes (64x2048, 32x1024 Darknet-21) are not stock configs
(SURVEY.md D5); the networks are fully convolutional, W must stay divisible by 16.
"""
from ..nets.Darknet import Darknet
from ..nets.SqueezeSegV2 import SqueezeSegV2
from ..configs import (SqueezeSegV2Config, SqueezeSegV2KittiConfig, SqueezeSegV2ConfigNuScenes,
                       Darknet53, Darknet21, Darknet53Kitti)

model_map = {
  "squeezesegv2": SqueezeSegV2,
  "darknet53": Darknet,
  "darknet21": Darknet,
}

config_map = {
  "squeezesegv2": SqueezeSegV2Config,
  "darknet53": Darknet53,
  "darknet21": Darknet21,
  "darknet53kitti": Darknet53Kitti,
  "squeezesegv2kitti": SqueezeSegV2KittiConfig,
  "squeezesegv2nuscenes": SqueezeSegV2ConfigNuScenes,
}


def load_model_config(model_name, config_name, height=None, width=None, **model_kw):
  config = config_map[config_name.lower()]()
  if height is not None:
    config.ZENITH_LEVEL = int(height)
  if width is not None:
    config.AZIMUTH_LEVEL = int(width)
  model = model_map[model_name.lower()](config, **model_kw)
  return config, model
