"""Bring a trained reference model over: SavedModel directory -> engine ``.npz``.

The reference keeps its weights in a TensorFlow SavedModel written by ``model.save()``
(train.py:110) and restores them with ``tf.keras.models.load_model`` (inference.py:39,
eval.py:40).  This script is the one step of the hand-over that needs TensorFlow: run it where
the model was trained, copy the ``.npz`` to the MI355X box, pass it to ``-p/--path_to_model``.

    python -m pclsegmentation_amd.tools.export_tf_weights \\
        --path_to_model ./output/model -m squeezesegv2 -c squeezesegv2 -o squeezesegv2.npz

It walks the restored object by the SAME attribute paths the reference's classes use
(``model.fire2.squeeze.kernel`` -> "fire2/squeeze/kernel"; nets/SqueezeSegV2.py:232-283,
nets/Darknet.py:187-260) — the list is ``nets/spec.py`` — so it does not depend on Keras' variable
naming.  TensorFlow is not installable in this repo's build image: the tree walk is covered by a
CPU test on a stand-in object tree, the TensorFlow call itself (two lines) is not.
"""
import argparse
import sys

import numpy as np

from ..nets import weights as W
from ..utils.args_loader import load_model_config

# Keras attribute names of the tensors behind each spec leaf
_LEAF = {"kernel": "kernel", "bias": "bias", "gamma": "gamma", "beta": "beta",
         "moving_mean": "moving_mean", "moving_variance": "moving_variance"}


def _to_numpy(v):
  return np.asarray(v.numpy() if hasattr(v, "numpy") else v, dtype=np.float32)


def collect_weights(model, spec):
  """{spec path: float32 array} read off ``model`` by attribute path; raises ValueError naming
  the first path that does not resolve or has the wrong shape."""
  out = {}
  for w in spec:
    *attrs, leaf = w.path.split("/")
    obj = model
    try:
      for a in attrs:
        obj = getattr(obj, a)
      out[w.path] = _to_numpy(getattr(obj, _LEAF[leaf]))
    except AttributeError as e:
      raise ValueError("model has no '%s' (%s)" % (w.path.replace("/", "."), e))
  W.check_weights(spec, out)
  return out


def main(argv=None):
  ap = argparse.ArgumentParser(description="Export a reference SavedModel to the engine's .npz")
  ap.add_argument("-p", "--path_to_model", required=True, help="SavedModel directory")
  ap.add_argument("-m", "--model", default="squeezesegv2")
  ap.add_argument("-c", "--config", default=None)
  ap.add_argument("-o", "--output", required=True)
  arg = ap.parse_args(argv)
  try:
    import tensorflow as tf
  except ImportError:
    sys.exit("this exporter runs where the model was trained: TensorFlow is required")
  config, model = load_model_config(arg.model, arg.config or arg.model)   # no GPU needed here
  keras_model = tf.keras.models.load_model(arg.path_to_model)
  model.set_weights(collect_weights(keras_model, model.weight_spec()))
  model.save(arg.output)                           # weights + arch + config, what load_model() reads
  print("wrote %d tensors (%d parameters) to %s"
        % (len(model.weights), sum(int(v.size) for v in model.weights.values()), arg.output))


if __name__ == "__main__":
  main()
