"""Helpers used by the inference CLI (reference: utils/util.py:99-100)."""


def normalize(x):
  return (x - x.min()) / (x.max() - x.min())
