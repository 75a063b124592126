#!/usr/bin/env python3
"""Generate tests/golden/configs.json from the reference's config functions.

Runs ONLY in the build container (needs /root/reference).  The reference configs
import ``easydict`` (absent from this image); they contain no arithmetic for the hot
path, only constants, so they are executed here with a minimal attribute-dict bound to
the name ``easydict.EasyDict`` and every field is dumped as data.  The JSON is the
fixture; this script is how it was made.

usage: python tests/golden/make_config_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True   # /root/reference is read-only by contract: no __pycache__ there
REF = "/root/reference/pcl_segmentation/configs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs.json")

# registry key -> (file, function)   (reference: utils/args_loader.py:42-49)
CONFIGS = {
  "squeezesegv2": ("SqueezeSegV2.py", "SqueezeSegV2Config"),
  "squeezesegv2kitti": ("SqueezeSegV2Kitti.py", "SqueezeSegV2KittiConfig"),
  "squeezesegv2nuscenes": ("SqueezeSegV2NuScenes.py", "SqueezeSegV2ConfigNuScenes"),
  "darknet53": ("Darknet53.py", "Darknet53"),
  "darknet21": ("Darknet21.py", "Darknet21"),
  "darknet53kitti": ("Darknet53Kitti.py", "Darknet53Kitti"),
}


class _AttrDict(dict):
  __getattr__ = dict.__getitem__
  __setattr__ = dict.__setitem__


def _jsonable(v):
  if isinstance(v, np.ndarray):
    return {"__ndarray__": True, "dtype": str(v.dtype), "shape": list(v.shape),
            "data": v.astype(np.float64).ravel().tolist()}
  if isinstance(v, dict):
    return {str(k): _jsonable(x) for k, x in v.items()}
  if isinstance(v, (list, tuple)):
    return [_jsonable(x) for x in v]
  if isinstance(v, (np.floating, np.integer)):
    return v.item()
  return v


def main():
  shim = types.ModuleType("easydict")
  shim.EasyDict = _AttrDict
  sys.modules["easydict"] = shim
  out = {}
  for key, (fname, func) in CONFIGS.items():
    spec = importlib.util.spec_from_file_location("refcfg_" + key, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mc = getattr(mod, func)()
    out[key] = {"function": func, "fields": _jsonable(dict(mc))}
  with open(OUT, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print("wrote", OUT)


if __name__ == "__main__":
  main()
