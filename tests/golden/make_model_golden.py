#!/usr/bin/env python3
"""Generate tests/golden/model_*.npz — known-answer vectors for whole-network parity.

The reference holds no golden outputs and TensorFlow cannot run here, so these vectors come
from this repo's float64 NumPy oracle (oracle/np_oracle.py; PARITY UNPINNED, see its
header) on the seeded synthetic weights (nets/weights.py, seed 4321) and seeded synthetic
scans (utils/synthetic.py, seed 1234), plus two REAL projected scans from the reference's
dataset_samples/sample_dataset/train (32x240x6 float64 files, stored as float32 data).

Each file holds: raw scans (float32 [n,H,W,5]), oracle logits (float32 of float64),
predictions (int32), and the top-1/top-2 logit margin used by the class-ID criterion.

usage: python tests/golden/make_model_golden.py   (needs /root/reference only for the real scans)
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import np_oracle as O  # noqa: E402
from pclsegmentation_amd import configs as C  # noqa: E402
from pclsegmentation_amd.nets.weights import synthetic_weights, spec_for_config  # noqa: E402
from pclsegmentation_amd.utils.synthetic import synthetic_scans  # noqa: E402

CASES = [
  # name, model, config fn, H, W, n, p_valid
  ("ssv2_32x240", "squeezesegv2", C.SqueezeSegV2Config, 32, 240, 2, 0.84),
  ("ssv2kitti_64x256", "squeezesegv2", C.SqueezeSegV2KittiConfig, 64, 256, 1, 0.78),
  ("darknet21_32x240", "darknet21", C.Darknet21, 32, 240, 1, 0.84),
  ("darknet53_32x240", "darknet53", C.Darknet53, 32, 240, 1, 0.84),
  ("darknet53kitti_16x64", "darknet53", C.Darknet53Kitti, 16, 64, 2, 0.78),
]


def run_case(model, mc, raw):
  lidar, mask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  w = synthetic_weights(spec_for_config(model, mc))
  prob, pred, logits = O.forward(model, w, lidar, mask, mc.CLASSES.index("None"),
                                 num_layers=mc.get("NUM_LAYERS"),
                                 output_stride=mc.get("OUTPUT_STRIDE", 16), dtype=np.float64)
  srt = np.sort(logits, axis=-1)
  margin = (srt[..., -1] - srt[..., -2]).astype(np.float32)
  return dict(raw=raw.astype(np.float32), logits=logits.astype(np.float32),
              preds=pred.astype(np.int32), margin=margin, mask=mask)


def main():
  for name, model, cfg, h, w, n, pv in CASES:
    mc = cfg()
    raw = synthetic_scans(n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, pv, seed=1234)
    out = run_case(model, mc, raw)
    np.savez_compressed(os.path.join(HERE, "model_%s.npz" % name), **out)
    print(name, out["logits"].shape, "logit std %.3f" % out["logits"].std())
  # two real scans of the reference's sample dataset (input fixtures; outputs from the oracle)
  files = sorted(glob.glob("/root/reference/dataset_samples/sample_dataset/train/*.npy"))[:2]
  if files:
    raw = np.stack([np.load(f).astype(np.float32)[:, :, :5] for f in files])
    mc = C.SqueezeSegV2Config()
    out = run_case("squeezesegv2", mc, raw)
    out["labels"] = np.stack([np.load(f)[:, :, 5].astype(np.int32) for f in files])
    np.savez_compressed(os.path.join(HERE, "model_ssv2_real_32x240.npz"), **out)
    print("real", out["logits"].shape)


if __name__ == "__main__":
  main()
