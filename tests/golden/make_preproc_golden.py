#!/usr/bin/env python3
"""Generate tests/golden/preproc_*.npz by RUNNING THE REFERENCE's own pre-processing —
``DataLoader.parse_sample`` (data_loader/data_loader.py:138-187, the twin of inference.py:47-62) — on real
scans of the reference's dataset_samples/.  Runs only in the build container (needs /root/reference).

This pins SURVEY.md §8 row a1 (normalise + mask + ``label[~mask] = None``) to the reference itself:
``parse_sample`` is pure NumPy.  The module needs TensorFlow only for (1) its ``import tensorflow as tf`` line,
(2) ``tf.io.gfile.glob`` in the constructor and (3) ``sample_path.numpy()`` — satisfied here by an otherwise
EMPTY stand-in module whose ``io.gfile.glob`` is ``glob.glob``, and a path object with ``.numpy()``.  The model
configs come from the reference's own config functions (executed as in make_config_golden.py: ``easydict`` is an
attribute-dict stand-in) and are cross-checked against tests/golden/configs.json.

Stored per case: ``sample`` = the reference's input file content ([H,W,6]: x, y, z, intensity, depth, label;
float32 when the file's float64 values are exactly representable, else float64), ``mean`` / ``std`` /
``none_index`` / ``cls_loss_weight`` = the config fields parse_sample reads, and its four outputs ``lidar``
(float32 [H,W,6]), ``mask`` (bool), ``label`` (int32), ``weight`` (float32).

usage: python tests/golden/make_preproc_golden.py
"""
import glob
import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True   # /root/reference is read-only by contract: no __pycache__ there
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# fixture name -> (dataset sample directory, indices into its sorted file list, config file, config function, registry key)
CASES = {
  "ika_train_32x240": ("sample_dataset/train", [0, 1, 17, 31], "SqueezeSegV2.py", "SqueezeSegV2Config", "squeezesegv2"),
  # this file's float64 values are NOT float32-representable: pins the `.astype(np.float32)` that precedes the maths
  "ika_val_32x240": ("sample_dataset/val", [0, 2], "SqueezeSegV2.py", "SqueezeSegV2Config", "squeezesegv2"),
  "kitti_val_64x1024": ("semantic_kitti/val", [0], "SqueezeSegV2Kitti.py", "SqueezeSegV2KittiConfig", "squeezesegv2kitti"),
  "nuscenes_val_32x1024": ("nuscenes/val", [0, 3], "SqueezeSegV2NuScenes.py", "SqueezeSegV2ConfigNuScenes", "squeezesegv2nuscenes"),
}


class _AttrDict(dict):
  __getattr__ = dict.__getitem__
  __setattr__ = dict.__setitem__


class _Path:
  """What tf.data hands to parse_sample: an object whose .numpy() is the file name."""
  def __init__(self, p):
    self._p = p

  def numpy(self):
    return self._p


def _load(path, name):
  spec = importlib.util.spec_from_file_location(name, path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def main():
  ed = types.ModuleType("easydict")
  ed.EasyDict = _AttrDict
  sys.modules["easydict"] = ed
  tf = types.ModuleType("tensorflow")          # empty but for the one function the constructor calls
  tf.io = types.SimpleNamespace(gfile=types.SimpleNamespace(glob=glob.glob))
  sys.modules["tensorflow"] = tf
  loader = _load(os.path.join(REF, "pcl_segmentation/data_loader/data_loader.py"), "ref_data_loader")
  dumped = json.load(open(os.path.join(HERE, "configs.json")))
  for name, (subdir, stems, cfg_file, cfg_func, key) in CASES.items():
    mc = getattr(_load(os.path.join(REF, "pcl_segmentation/configs", cfg_file), "refcfg_" + key), cfg_func)()
    for field in ("INPUT_MEAN", "INPUT_STD"):
      assert np.array_equal(np.asarray(mc[field]).ravel(), dumped[key]["fields"][field]["data"]), field
    root, split = os.path.split(os.path.join(REF, "dataset_samples", subdir))
    dl = loader.DataLoader(split, root, mc)
    assert len(dl._sample_pathes) > 0
    samples, outs = [], [[], [], [], []]
    listing = sorted(os.listdir(os.path.join(root, split)))
    stems = [listing[i] for i in stems]
    for stem in stems:
      path = os.path.join(root, split, stem)
      raw = np.load(path)
      lossless = np.array_equal(raw.astype(np.float32).astype(np.float64), raw)
      samples.append(raw.astype(np.float32) if lossless else raw)
      for o, v in zip(outs, dl.parse_sample(_Path(path))):
        o.append(v)
    dt = np.float32 if all(s.dtype == np.float32 for s in samples) else np.float64
    lidar, mask, label, weight = (np.stack(o) for o in outs)
    assert lidar.dtype == np.float32 and mask.dtype == bool and label.dtype == np.int32 and weight.dtype == np.float32
    np.savez_compressed(os.path.join(HERE, "preproc_%s.npz" % name),
                        sample=np.stack(samples).astype(dt), files=np.array([os.path.join(subdir, s) for s in stems]),
                        config=key, mean=np.asarray(mc.INPUT_MEAN, np.float64).ravel(), std=np.asarray(mc.INPUT_STD, np.float64).ravel(),
                        none_index=mc.CLASSES.index("None"), num_class=mc.NUM_CLASS,
                        cls_loss_weight=np.asarray(mc.CLS_LOSS_WEIGHT, np.float64),
                        lidar=lidar, mask=mask, label=label, weight=weight)
    print("%-22s %s sample %s  valid %.3f  %d KB" % (name, lidar.shape, dt.__name__, mask.mean(),
          os.path.getsize(os.path.join(HERE, "preproc_%s.npz" % name)) // 1024))


if __name__ == "__main__":
  main()
