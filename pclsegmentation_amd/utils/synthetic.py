"""Seeded synthetic range images with the statistics of real projected scans.

SURVEY.md §8(d): ``valid ~ Bernoulli(p)`` per pixel; for valid pixels
``x, y, z, intensity ~ N(mean_c, std_c)`` with the config's INPUT_MEAN/STD and
``depth = max(||(x,y,z)||, 0.05)``; invalid pixels are all-zero (depth 0 => mask false,
as the reference's converters write them: dataset_convert/semantic_kitti.py:162-165).
"""
import numpy as np

# measured valid-pixel rates of the reference's sample scans
VALID_RATE = {"kitti": 0.78, "nuscenes": 0.59, "ika": 0.84}


def synthetic_scans(n, h, w, mean, std, p_valid=0.78, seed=1234):
  """-> float32 [n, h, w, 5] raw scans (x, y, z, intensity, depth)."""
  rng = np.random.default_rng(seed)
  mean = np.asarray(mean, np.float64).reshape(5)
  std = np.asarray(std, np.float64).reshape(5)
  valid = rng.random((n, h, w)) < p_valid
  feat = rng.normal(mean[:4], std[:4], size=(n, h, w, 4))
  depth = np.maximum(np.sqrt((feat[..., :3] ** 2).sum(-1)), 0.05)
  scans = np.concatenate([feat, depth[..., None]], axis=-1)
  scans[~valid] = 0.0
  return scans.astype(np.float32)


def synthetic_scan_range(lo, hi, h, w, mean, std, p_valid=0.78, seed=1234, chunk=8):
  """Scans [lo, hi) of ONE seeded job batch of any length, generated in chunks of ``chunk`` scans
  (chunk c is ``synthetic_scans(chunk, ..., seed=seed + c)``): a rank of a sharded job materialises only
  the scans it owns, and every rank — and the single-process check — sees the same scan i."""
  if hi <= lo:
    return np.zeros((0, h, w, 5), np.float32)
  parts = []
  for c in range(lo // chunk, (hi - 1) // chunk + 1):
    s = synthetic_scans(chunk, h, w, mean, std, p_valid, seed=seed + c)
    parts.append(s[max(lo - c * chunk, 0):min(hi - c * chunk, chunk)])
  return np.concatenate(parts, axis=0)
