"""pclsegmentation_amd — MI355X-native forward-pass engine for SqueezeSegV2 / Darknet-21/53
LiDAR range-image segmentation, behind the reference's own Python surface
(configs, load_model_config, ``model([lidar, mask]) -> (probabilities, predictions)``,
the inference CLI).  Everything numerical runs in libpclseg.so (hand-written HIP, gfx950)."""
import json as _json

import numpy as _np

from . import configs
from .utils.args_loader import load_model_config, model_map, config_map
from .nets import SqueezeSegV2, Darknet

__all__ = ["configs", "load_model_config", "load_model", "model_map", "config_map",
           "SqueezeSegV2", "Darknet"]


def load_model(path, model_name=None, config_name=None, **model_kw):
  """Counterpart of ``tf.keras.models.load_model`` (reference: inference.py:39, eval.py:40).
  ``path`` is either one of the engine's own ``.npz`` model files (written by ``model.save`` or by
  tools/export_tf_weights.py) or a reference SavedModel DIRECTORY, whose tensor bundle is read
  without TensorFlow (savedmodel.py); a SavedModel does not record which config it was trained
  with, so ``model_name`` (and optionally ``config_name``) must be given for it."""
  from .nets import weights as _w
  from . import savedmodel as _sm
  if _sm.is_savedmodel_dir(path):
    if not model_name:
      raise ValueError("%s is a SavedModel directory: pass model_name (and config_name)" % path)
    _, model = load_model_config(model_name, config_name or model_name, **model_kw)
    model.set_weights(_sm.load_savedmodel_weights(path, model.weight_spec()))
    return model
  weights, meta = _w.load_weights(path)
  if "arch" not in meta or "config_json" not in meta:
    raise ValueError("%s is not a pclsegmentation_amd model file (no arch/config metadata)" % path)
  cfg = configs.ModelConfig(_json.loads(str(meta["config_json"])))
  for k in ("INPUT_MEAN", "INPUT_STD", "CLS_COLOR_MAP", "CLS_LOSS_WEIGHT"):
    if k in cfg:
      cfg[k] = _np.array(cfg[k])
  arch = str(meta["arch"])
  model = model_map[arch](cfg, **model_kw)
  model.set_weights(weights)
  return model
