"""Weight sets: deterministic synthetic initialisation and ``.npz`` I/O.

The reference ships no trained weights and its SavedModel format needs TensorFlow
(reference: train.py:60, inference.py:39), so the engine's own weight file is a flat
``.npz`` keyed by Keras attribute path (see nets/spec.py) with Keras tensor layouts.

``synthetic_weights`` is the seeded recipe used by the benchmark, the golden fixtures
and the parity tests: He-scaled kernels and near-identity BatchNorm
statistics, so that activations stay O(1) through the deepest network and an absolute
logit tolerance of 1e-3 is meaningful.
"""
import numpy as np

from .spec import weight_spec


def synthetic_weights(spec, seed=4321, residual_gain=0.1):
  """Draw every tensor of ``spec`` in order from ``default_rng(seed)``.

  kernels ~ N(0, sqrt(2/fan_in)); biases ~ N(0, 0.05); gamma ~ U(0.8, 1.2);
  beta, moving_mean ~ N(0, 0.05); moving_variance ~ U(0.8, 1.2).

  ``residual_gain`` scales the gamma of the LAST BatchNorm in every Darknet BasicBlock
  (``.../bn2/gamma``).  The block computes ``x + f(x)``; with a unit-gain branch the
  activation variance doubles per block (2**26 over Darknet-53) and logits reach 1e4,
  where float32 itself cannot hold 1e-3.  A gain of 0.1 keeps the logit std at 4-6 for all three nets.
  """
  rng = np.random.default_rng(seed)
  out = {}
  for w in spec:
    if w.kind in ("conv", "deconv"):
      v = rng.normal(0.0, np.sqrt(2.0 / w.fan_in), size=w.shape)
    elif w.kind in ("bias", "beta", "mean"):
      v = rng.normal(0.0, 0.05, size=w.shape)
    elif w.kind == "gamma":
      v = rng.uniform(0.8, 1.2, size=w.shape)
      if w.path.endswith("/bn2/gamma"):
        v = v * residual_gain
    elif w.kind == "var":
      v = rng.uniform(0.8, 1.2, size=w.shape)
    else:
      raise ValueError(w.kind)
    out[w.path] = v.astype(np.float32)
  return out


def keras_default_weights(spec, seed=0):
  """What a freshly constructed Keras model holds: glorot-uniform kernels, zero biases,
  gamma=1, beta=0, moving_mean=0, moving_variance=1."""
  rng = np.random.default_rng(seed)
  out = {}
  for w in spec:
    if w.kind == "conv":
      kh, kw, cin, cout = w.shape
      lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
      v = rng.uniform(-lim, lim, size=w.shape)
    elif w.kind == "deconv":
      kh, kw, cout, cin = w.shape
      lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
      v = rng.uniform(-lim, lim, size=w.shape)
    elif w.kind in ("gamma", "var"):
      v = np.ones(w.shape)
    else:
      v = np.zeros(w.shape)
    out[w.path] = v.astype(np.float32)
  return out


def check_weights(spec, weights):
  """Raise ValueError naming the first missing / mis-shaped tensor."""
  for w in spec:
    if w.path not in weights:
      raise ValueError("missing weight '%s' %s" % (w.path, (w.shape,)))
    got = tuple(np.shape(weights[w.path]))
    if got != tuple(w.shape):
      raise ValueError("weight '%s' has shape %s, expected %s" % (w.path, got, tuple(w.shape)))


def save_weights(path, weights, meta=None):
  """Write a weight set (+ optional metadata such as arch / NUM_CLASS) as .npz."""
  arrays = {k.replace("/", "|"): np.asarray(v, dtype=np.float32) for k, v in weights.items()}
  if meta:
    for k, v in meta.items():
      arrays["__meta__" + k] = np.asarray(v)
  np.savez(path, **arrays)


def load_weights(path):
  """Inverse of save_weights -> (weights dict, meta dict)."""
  weights, meta = {}, {}
  with np.load(path, allow_pickle=False) as z:
    for k in z.files:
      if k.startswith("__meta__"):
        v = z[k]
        meta[k[len("__meta__"):]] = v.item() if v.shape == () else v
      else:
        weights[k.replace("|", "/")] = z[k]
  return weights, meta


def spec_for_config(model_name, mc):
  """Weight inventory of ``model_name`` under config ``mc``."""
  name = model_name.lower()
  if name == "squeezesegv2":
    return weight_spec("squeezesegv2", mc.NUM_CLASS)
  return weight_spec("darknet", mc.NUM_CLASS, num_layers=mc.NUM_LAYERS,
                     output_stride=mc.OUTPUT_STRIDE)
