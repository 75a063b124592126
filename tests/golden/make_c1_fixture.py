#!/usr/bin/env python3
"""Generate tests/golden/c1_sample_dataset_train_32x240.npz — the INPUTS of BASELINE.json configs[0]:
all 32 projected scans of the reference's dataset_samples/sample_dataset/train (files are (32,240,6)
float64: x, y, z, intensity, depth, label), stored as float32 [32,32,240,5] raw scans + int8 labels +
the file stems (the CLI names its outputs after them, inference.py:64-66).

Data only — inputs the reference's own repository holds; expected outputs come from the float64 oracle
at test time (the reference ships no outputs and no trained weights).

usage: python tests/golden/make_c1_fixture.py      (needs /root/reference)
"""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
  files = sorted(glob.glob("/root/reference/dataset_samples/sample_dataset/train/*.npy"))
  assert len(files) == 32, len(files)
  data = [np.load(f) for f in files]
  assert all(d.shape == (32, 240, 6) for d in data)
  raw = np.stack([d[:, :, :5].astype(np.float32) for d in data])
  labels = np.stack([d[:, :, 5] for d in data])
  assert np.array_equal(labels, labels.astype(np.int8))
  np.savez_compressed(os.path.join(HERE, "c1_sample_dataset_train_32x240.npz"), raw=raw,
                      labels=labels.astype(np.int8), stems=np.array([os.path.basename(f)[:-4] for f in files]))
  print(raw.shape, labels.min(), labels.max(), os.path.getsize(os.path.join(HERE, "c1_sample_dataset_train_32x240.npz")))


if __name__ == "__main__":
  main()
