"""Base class of the segmentation networks: the host-side mirror of the reference's Keras
model object (reference: nets/SegmentationNetwork.py:28-69, :133-136).

Only the inference contract is mirrored — ``model([lidar, mask]) -> (probabilities,
predictions)`` with ``training=False`` semantics (BatchNorm moving statistics, dropout
off).  Loss, train_step/test_step and metrics (:71-131) are training-only and out of scope.

All arithmetic happens in libpclseg.so (HIP, gfx950) through ``engine.Engine``; nothing in
this class computes a network operation in Python, and without a GPU it raises.
"""
import json

import numpy as np

from .. import engine as _engine
from . import weights as _weights


class EagerArray(np.ndarray):
  """ndarray with the ``.numpy()`` accessor callers of the reference use (inference.py:78)."""

  def numpy(self):
    return np.asarray(self)


def _eager(a):
  return np.asarray(a).view(EagerArray)


def _is_torch(x):
  return type(x).__module__.startswith("torch")


class PCLSegmentationNetwork:
  ARCH = None  # set by subclasses: "squeezesegv2" | "darknet"

  def __init__(self, mc, device=0, micro_batch=0):
    self.mc = mc
    self.NUM_CLASS = mc.NUM_CLASS
    self.BATCH_SIZE = mc.BATCH_SIZE
    self.ZENITH_LEVEL = mc.ZENITH_LEVEL
    self.AZIMUTH_LEVEL = mc.AZIMUTH_LEVEL
    self.NUM_FEATURES = mc.NUM_FEATURES
    self.CLASSES = mc.CLASSES
    self.CLS_COLOR_MAP = mc.CLS_COLOR_MAP
    self.device = device
    self.micro_batch = micro_batch
    self.weights = None
    self._engines = {}

  # ---- architecture hooks
  def arch_name(self):
    raise NotImplementedError("Method should be called in child class!")

  def weight_spec(self):
    return _weights.spec_for_config(self.arch_name(), self.mc)

  # ---- weights
  def set_weights(self, weights):
    """Bind a weight set: dict Keras-path -> array in Keras layout."""
    _weights.check_weights(self.weight_spec(), weights)
    self.weights = {w.path: np.asarray(weights[w.path], np.float32) for w in self.weight_spec()}
    self._drop_engines()
    return self

  def init_weights(self, seed=4321):
    """Deterministic synthetic weights (there are no trained weights to ship)."""
    return self.set_weights(_weights.synthetic_weights(self.weight_spec(), seed))

  def load_weights(self, path):
    w, _ = _weights.load_weights(path)
    return self.set_weights(w)

  def save(self, path):
    """Counterpart of model.save (reference: train.py:60): one .npz holding the Keras-path
    keyed tensors plus what is needed to rebuild the model."""
    if self.weights is None:
      raise RuntimeError("model has no weights to save")
    cfg = {}
    for k, v in self.mc.items():
      cfg[k] = v.tolist() if isinstance(v, np.ndarray) else v
    _weights.save_weights(path, self.weights,
                          meta={"arch": self.arch_name(), "config_json": json.dumps(cfg)})

  # ---- engines (one per input size: the graph is fully convolutional, reference D5)
  def _drop_engines(self):
    for e in self._engines.values():
      e.close()
    self._engines = {}

  def engine(self, height=None, width=None, flags=0):
    h = int(height or self.ZENITH_LEVEL)
    w = int(width or self.AZIMUTH_LEVEL)
    key = (h, w, flags)
    if key not in self._engines:
      if self.weights is None:
        raise RuntimeError("model has no weights: call set_weights / init_weights / load_weights")
      eng = _engine.Engine(self.engine_desc(h, w, flags))
      eng.set_weights(self.weights)
      eng.finalize()
      self._engines[key] = eng
    return self._engines[key]

  def engine_desc(self, height=None, width=None, flags=0):
    """pclseg_desc of this model at a given range-image size (what pclseg_create takes)."""
    mc = self.mc
    return _engine.make_desc(self.arch_name(), int(height or self.ZENITH_LEVEL), int(width or self.AZIMUTH_LEVEL),
                             mc.NUM_CLASS, mc.CLASSES.index("None"), mc.INPUT_MEAN, mc.INPUT_STD,
                             output_stride=mc.get("OUTPUT_STRIDE", 16), device=self.device,
                             micro_batch=self.micro_batch, flags=flags)

  def adopt_engine(self, eng, height, width, flags=0):
    """Register an engine whose parameters arrived as a packed blob (distributed.broadcast_engine):
    this rank never holds the Keras tensors."""
    self._engines[(int(height), int(width), flags)] = eng
    return eng

  # ---- the model call
  def call(self, inputs, training=False, mask=None, return_probabilities=True):
    if training:
      raise NotImplementedError("this engine implements the inference path only")
    lidar_input, lidar_mask = inputs[0], inputs[1]
    if _is_torch(lidar_input) and _engine.on_device(lidar_input):     # (a CPU tensor takes the host path below)
      return self._call_device(lidar_input, lidar_mask, return_probabilities)
    lidar = np.ascontiguousarray(np.asarray(lidar_input), dtype=np.float32)  # Keras casts to f32
    msk = np.asarray(lidar_mask)
    if lidar.ndim != 4 or lidar.shape[-1] != self.NUM_FEATURES:
      raise ValueError("lidar must have shape [N,H,W,%d], got %s" % (self.NUM_FEATURES, lidar.shape))
    msk = msk.reshape(msk.shape[:3]) if msk.ndim == 4 else msk
    if msk.shape != lidar.shape[:3]:
      raise ValueError("mask shape %s does not match lidar %s" % (msk.shape, lidar.shape))
    msk = np.ascontiguousarray(msk.astype(np.uint8))
    n, h, w, _ = lidar.shape
    eng = self.engine(h, w)
    preds = np.empty((n, h, w), np.int32)
    probs = np.empty((n, h, w, self.NUM_CLASS), np.float32) if return_probabilities else None
    eng.forward(lidar, msk, n, preds, probs, None, mem=_engine.MEM_HOST)
    return (_eager(probs) if probs is not None else None), _eager(preds)

  def _call_device(self, lidar, mask, return_probabilities):
    import torch
    if lidar.dtype != torch.float32:
      lidar = lidar.float()
    lidar = lidar.contiguous()
    mask = mask.reshape(lidar.shape[:3]).to(torch.uint8).contiguous()
    n, h, w, c = lidar.shape
    if c != self.NUM_FEATURES:
      raise ValueError("lidar must have shape [N,H,W,%d]" % self.NUM_FEATURES)
    eng = self.engine(h, w)
    eng.set_stream(_engine.stream_handle(lidar.device))
    preds = torch.empty((n, h, w), dtype=torch.int32, device=lidar.device)
    probs = (torch.empty((n, h, w, self.NUM_CLASS), dtype=torch.float32, device=lidar.device)
             if return_probabilities else None)
    eng.forward(lidar, mask, n, preds, probs, None, mem=_engine.MEM_DEVICE)
    return probs, preds

  def __call__(self, inputs, training=False, mask=None, **kw):
    return self.call(inputs, training=training, mask=mask, **kw)

  def predict_step(self, data):
    """reference: nets/SegmentationNetwork.py:133-136."""
    (lidar_input, lidar_mask), _, _ = data
    return self.call([lidar_input, lidar_mask], training=False)

  def predict_raw(self, scans, return_mask=False):
    """Raw scans [N,H,W,5] (x,y,z,intensity,depth) -> predictions, with the reference's
    caller-side pre-processing (inference.py:50-62) done on the device."""
    if _is_torch(scans) and _engine.on_device(scans):
      import torch
      scans = scans.float().contiguous()
      n, h, w, _ = scans.shape
      eng = self.engine(h, w)
      eng.set_stream(_engine.stream_handle(scans.device))
      preds = torch.empty((n, h, w), dtype=torch.int32, device=scans.device)
      mask = torch.empty((n, h, w), dtype=torch.uint8, device=scans.device) if return_mask else None
      eng.forward_raw(scans, n, preds, None, None, mask, mem=_engine.MEM_DEVICE)
      return (preds, mask) if return_mask else preds
    scans = np.ascontiguousarray(np.asarray(scans)[..., :5], dtype=np.float32)
    n, h, w, _ = scans.shape
    eng = self.engine(h, w)
    preds = np.empty((n, h, w), np.int32)
    mask = np.empty((n, h, w), np.uint8) if return_mask else None
    eng.forward_raw(scans, n, preds, None, None, mask, mem=_engine.MEM_HOST)
    return (_eager(preds), mask.astype(bool)) if return_mask else _eager(preds)

  def get_config(self):
    return {"mc": self.mc}

  @classmethod
  def from_config(cls, config):
    return cls(**config)
