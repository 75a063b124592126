"""Operator-level parity on the MI355X: each HIP kernel, called through the C ABI's
single-operator entry points, against the NumPy oracle on the same seeded inputs.

Covers the TF-semantics edge cases the reference relies on:
asymmetric SAME padding under stride 2, odd/even widths, tiles that do not divide the
image, the transposed-conv index map, -inf pool borders, argmax ties, masked pixels.
Tolerances are float32 round-off (the kernels compute in exact float32 on the matrix
cores; only the summation order differs from the oracle's).
"""
import zlib

import numpy as np
import pytest

from oracle import np_oracle as O
from pclsegmentation_amd import engine as E


pytestmark = pytest.mark.gpu


def dev(a, cuda):
  import torch
  return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def rnd(rng, *shape):
  return rng.standard_normal(shape).astype(np.float32)


def bn_params(rng, c):
  return (rng.uniform(0.8, 1.2, c).astype(np.float32), rng.normal(0, 0.05, c).astype(np.float32),
          rng.normal(0, 0.05, c).astype(np.float32), rng.uniform(0.8, 1.2, c).astype(np.float32))


ACTS = {"none": lambda x: x, "relu": O.relu, "leaky": O.leaky_relu, "sigmoid": O.sigmoid}


def test_normalize_bit_exact(cuda):
  import torch
  rng = np.random.default_rng(7)
  mean = [24.810, 0.819, 0.000, 16.303, 25.436]
  std = [30.335, 7.807, 2.058, 25.208, 30.897]
  raw = (rng.standard_normal((3, 32, 240, 5)) * 20).astype(np.float32)
  raw[..., 4] = np.abs(raw[..., 4])
  raw[rng.random((3, 32, 240)) < 0.3] = 0.0           # empty pixels
  raw[0, 0, 0] = [1, 2, 3, 4, -1.0]                    # negative depth is invalid (strict >)
  raw[0, 0, 1] = [1, 2, 3, 4, 0.0]
  want, wmask = O.normalize_and_mask(raw, mean, std)
  out = torch.empty((3, 32, 240, 6), dtype=torch.float32, device=cuda)
  mask = torch.empty((3, 32, 240), dtype=torch.uint8, device=cuda)
  E.op_normalize(dev(raw, cuda), 3, 32, 240, mean, std, out, mask)
  assert np.array_equal(mask.cpu().numpy().astype(bool), wmask)
  assert np.array_equal(out.cpu().numpy(), want.astype(np.float32))  # float64 math, one rounding


CONV_CASES = [
  # n, h, w, cin, cout, k, stride, act, bias, bn, residual
  (1, 8, 16, 16, 16, 3, 1, "none", False, False, False),
  (2, 8, 32, 16, 64, 3, 1, "relu", True, True, False),
  (1, 13, 37, 32, 48, 3, 1, "leaky", False, True, True),     # ragged tiles, 3 cout tiles
  (1, 5, 15, 64, 128, 3, 1, "relu", True, True, False),      # W = 15 (240/16)
  (1, 16, 64, 8, 64, 3, 2, "relu", True, True, False),       # stem-like, stride 2, even W
  (1, 8, 34, 32, 64, 3, 2, "leaky", False, True, False),     # stride 2, W % 4 == 2
  (1, 8, 33, 16, 32, 3, 2, "none", True, False, False),      # stride 2, odd W: pad 1/1
  (1, 4, 4, 4, 4, 3, 2, "none", False, False, False),        # KAT-sized
  (1, 4, 6, 4, 4, 3, 2, "none", False, False, False),
  (2, 8, 40, 64, 16, 1, 1, "relu", True, True, False),       # squeeze
  (1, 7, 19, 128, 8, 1, 1, "relu", True, True, False),       # CAM squeeze
  (1, 7, 19, 8, 128, 1, 1, "sigmoid", True, True, False),    # CAM excitation
  (1, 8, 16, 384, 48, 1, 1, "relu", True, True, False),      # multi-chunk Cin, 3 cout tiles
  (1, 8, 16, 48, 192, 3, 1, "relu", True, True, True),       # partial last chunk (48 = 32+16)
  (1, 8, 16, 256, 512, 3, 1, "leaky", False, True, True),    # darknet-sized
  (1, 3, 130, 4, 20, 1, 1, "none", True, False, False),      # cout 20 -> padded tile
]
# Darknet's wide 1x1 layers (BasicBlock / decoder-block conv1): flat pixels, 128-pixel tiles (ragged last tile),
# 2 .. 16 channel chunks, one to four cout groups — the shapes conv1x1_wide_kernel takes
WIDE_1X1_CASES = [
  (1, 8, 40, 256, 128, 1, 1, "leaky", False, True, False),   # enc3-like: 8 cout tiles, 320 px = 2.5 tiles
  (2, 4, 33, 128, 256, 1, 1, "leaky", False, True, False),   # dec3-like: 2 chunks, 16 cout tiles, 264 px
  (1, 4, 96, 1024, 512, 1, 1, "leaky", False, True, False),  # enc5-like: 16 chunks, 2 cout groups, 3 tiles
  (1, 2, 64, 512, 1024, 1, 1, "leaky", True, True, False),   # dec5-like: 4 cout groups, exactly one tile
]


MATHS = ["f16x3", "f32"]


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d(cuda, case, math):
  _conv2d_case(cuda, case, math)


@pytest.mark.first_hw_run
@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("case", WIDE_1X1_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d_wide_1x1(cuda, case, math):
  _conv2d_case(cuda, case, math)


def _conv2d_case(cuda, case, math):
  import torch
  n, h, w, cin, cout, k, s, act, use_bias, use_bn, use_res = case
  rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
  x = rnd(rng, n, h, w, cin)
  kern = (rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
  bias = rng.normal(0, 0.05, cout).astype(np.float32) if use_bias else None
  bn = bn_params(rng, cout) if use_bn else None
  y64 = O.conv2d(x.astype(np.float64), kern.astype(np.float64),
                 None if bias is None else bias.astype(np.float64), s)
  if bn is not None:
    y64 = O.batch_norm(y64, *[b.astype(np.float64) for b in bn])
  y64 = ACTS[act](y64)
  res = rnd(rng, *y64.shape) if use_res else None
  if res is not None:
    y64 = y64 + res
  y = torch.full(y64.shape, float("nan"), dtype=torch.float32, device=cuda)
  E.op_conv2d(dev(x, cuda), n, h, w, cin, kern, s, bias, bn, act,
              None if res is None else dev(res, cuda), y, math)
  got = y.cpu().numpy()
  assert np.isfinite(got).all(), "kernel left outputs unwritten"
  err = np.abs(got - y64).max()
  tol = 2e-5 * max(1.0, np.abs(y64).max())
  _log_margin("conv2d", case, math, err, tol)
  assert err <= tol, err


def _log_margin(kind, case, math, err, tol):
  """PCLSEG_TOL_LOG=<file>: one line per case with err / tol (how much of the tolerance a build uses; scripts/tol_margin.py)."""
  import os
  path = os.environ.get("PCLSEG_TOL_LOG")
  if path:
    with open(path, "a") as fh:
      fh.write("%s\t%s\t%s\t%.3e\t%.3e\t%.4f\n" % (kind, "x".join(str(v) for v in case), math, err, tol, err / tol))


def _fuzz_cases(count, seed):
  """Seeded random convolution shapes over the whole dispatch space of pclseg_op_conv2d: channel counts on both
  sides of every chunk / cout-tile boundary, 1x1 and 3x3, both strides, ragged and tiny images, every epilogue."""
  rng = np.random.default_rng(seed)
  cins = [4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 320, 512, 1024]
  couts = [4, 8, 16, 20, 32, 48, 64, 80, 96, 128, 192, 256, 512, 1024]
  out = []
  while len(out) < count:
    k = int(rng.choice([1, 3]))
    s = int(rng.choice([1, 2])) if k == 3 else 1
    cin, cout = int(rng.choice(cins)), int(rng.choice(couts))
    n, h, w = int(rng.integers(1, 3)), int(rng.integers(1, 11)), int(rng.integers(1, 141))
    if n * h * w * k * k * cin * cout > 3e8:
      continue
    out.append((n, h, w, cin, cout, k, s, str(rng.choice(["none", "relu", "leaky", "sigmoid"])), bool(rng.integers(2)),
                bool(rng.integers(2)), bool(rng.integers(2))))
  return out


@pytest.mark.first_hw_run
@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("case", _fuzz_cases(160, 20261004), ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d_fuzz(cuda, case, math):
  """Every shape is either computed to the oracle's values or rejected with a shape error — never wrong, never a
  crash, never a partly written output (the kernels behind the operator entry point are picked by a dispatch table
  of tile shapes and epilogues of which the three networks use a fraction).

  Tolerance (verdict r5 #8).  2e-5 x max(1, max|y|) is not a calibration on a few hardware cases: with K = k*k*cin products per
  output, float32 accumulation contributes about sqrt(R) x 2^-24 x |y| (R = roundings of the accumulator: K in exact mode and on
  the simulator, which rounds after every k; at most 3 K / 32 block sums on the device's f16 MFMA), i.e. <= 6e-6 |y| (1 sigma) at
  the largest K here (9216); the dropped lo x lo term and the activation split (absolute 2^-25 below 1/8) are two orders
  smaller.  Measured on the simulator's pessimistic rounding order the worst of the 448 convolution cases uses 34 % of the
  tolerance (profiles/r06_sim_tolerance_margin.txt, K = 9216); the 16 cases round 3 ran on an MI355X passed with the same
  bound."""
  try:
    _conv2d_case(cuda, case, math)
  except ValueError as e:
    assert "pclseg" in str(e), e


def test_conv2d_stride2_known_answer(cuda):
  """row [1,2,3,4], k=3, s=2, kernel [1,10,100] -> [321, 43]
  (even W pads right only)."""
  import torch
  x = np.zeros((1, 1, 4, 4), np.float32)
  x[0, 0, :, 0] = [1, 2, 3, 4]
  kern = np.zeros((3, 3, 4, 4), np.float32)
  kern[1, :, 0, 0] = [1, 10, 100]
  y = torch.zeros((1, 1, 2, 4), dtype=torch.float32, device=cuda)
  for math in MATHS:
    E.op_conv2d(dev(x, cuda), 1, 1, 4, 4, kern, 2, None, None, "none", None, y, math)
    assert y.cpu().numpy()[0, 0, :, 0].tolist() == [321.0, 43.0]


@pytest.mark.parametrize("case", [(1, 8, 16, 16, 16, "relu", False), (2, 5, 15, 64, 64, "relu", False),
                                  (1, 9, 40, 32, 32, "leaky", True), (1, 8, 8, 512, 256, "leaky", True)],
                         ids=str)
@pytest.mark.parametrize("math", MATHS)
def test_conv2d_transpose(cuda, case, math):
  _conv2d_transpose_case(cuda, case, math)


def _transpose_fuzz_cases(count, seed):
  rng = np.random.default_rng(seed)
  out = []
  while len(out) < count:
    cin = int(rng.choice([8, 16, 32, 48, 64, 96, 128, 256, 512, 1024]))
    cout = int(rng.choice([8, 16, 32, 48, 64, 96, 128, 256, 512]))
    n, h, w = int(rng.integers(1, 3)), int(rng.integers(1, 10)), int(rng.integers(1, 71))
    if n * h * w * 4 * cin * cout > 3e8:
      continue
    out.append((n, h, w, cin, cout, str(rng.choice(["none", "relu", "leaky"])), bool(rng.integers(2))))
  return out


@pytest.mark.first_hw_run
@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("case", _transpose_fuzz_cases(40, 20261005), ids=str)
def test_conv2d_transpose_fuzz(cuda, case, math):
  """Conv2DTranspose (1,4)/(1,2) over the dispatch space of its two parity sub-convolutions (FIREUP and the Darknet
  decoder use four shapes of it): computed to the oracle's values or rejected with a shape error."""
  try:
    _conv2d_transpose_case(cuda, case, math)
  except ValueError as e:
    assert "pclseg" in str(e), e


def _conv2d_transpose_case(cuda, case, math):
  import torch
  n, h, w, cin, cout, act, use_bn = case
  rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
  x = rnd(rng, n, h, w, cin)
  kern = (rng.standard_normal((1, 4, cout, cin)) * np.sqrt(1.0 / cin)).astype(np.float32)
  bias = rng.normal(0, 0.05, cout).astype(np.float32)
  bn = bn_params(rng, cout) if use_bn else None
  y64 = O.conv2d_transpose_1x4_s2(x.astype(np.float64), kern.astype(np.float64), bias.astype(np.float64))
  if bn is not None:
    y64 = O.batch_norm(y64, *[b.astype(np.float64) for b in bn])
  y64 = ACTS[act](y64)
  y = torch.full(y64.shape, float("nan"), dtype=torch.float32, device=cuda)
  E.op_conv2d_transpose(dev(x, cuda), n, h, w, cin, kern, bias, bn, act, y, math)
  got = y.cpu().numpy()
  assert np.isfinite(got).all()
  err, tol = np.abs(got - y64).max(), 2e-5 * max(1.0, np.abs(y64).max())
  _log_margin("conv2d_transpose", case, math, err, tol)
  assert err <= tol, err


def test_conv2d_transpose_impulse_known_answer(cuda):
  """K=[a,b,c,d]; x=[1,0,0] -> [b,c,d,0,0,0]; x=[0,1,0] -> [0,a,b,c,d,0]."""
  import torch
  a, b, c, d = 2.0, 3.0, 5.0, 7.0
  kern = np.zeros((1, 4, 4, 4), np.float32)
  kern[0, :, 0, 0] = [a, b, c, d]
  for pos, want in ((0, [b, c, d, 0, 0, 0]), (1, [0, a, b, c, d, 0])):
    x = np.zeros((1, 1, 3, 4), np.float32)
    x[0, 0, pos, 0] = 1.0
    y = torch.zeros((1, 1, 6, 4), dtype=torch.float32, device=cuda)
    for math in MATHS:
      E.op_conv2d_transpose(dev(x, cuda), 1, 1, 3, 4, kern, np.zeros(4, np.float32), None, "none", y, math)
      assert y.cpu().numpy()[0, 0, :, 0].tolist() == want


@pytest.mark.parametrize("case", [(2, 9, 21, 8, 7, 1), (1, 64, 32, 64, 7, 1), (1, 3, 4, 4, 7, 1),
                                  (2, 8, 32, 16, 3, 2), (1, 5, 33, 4, 3, 2), (1, 1, 4, 4, 3, 2)], ids=str)
def test_max_pool_bit_exact(cuda, case):
  import torch
  n, h, w, c, k, s = case
  rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
  x = rnd(rng, n, h, w, c) - 3.0      # mostly negative: a zero-padded pool would be wrong
  want = O.max_pool(x, k, s)
  y = torch.full(want.shape, float("nan"), dtype=torch.float32, device=cuda)
  E.op_max_pool(dev(x, cuda), n, h, w, c, k, s, y)
  assert np.array_equal(y.cpu().numpy(), want)


def test_max_pool_known_answer(cuda):
  """max-pool 3 s2 of [5,1,2,9] -> [5, 9]."""
  import torch
  x = np.zeros((1, 1, 4, 4), np.float32)
  x[0, 0, :, 0] = [5, 1, 2, 9]
  y = torch.zeros((1, 1, 2, 4), dtype=torch.float32, device=cuda)
  E.op_max_pool(dev(x, cuda), 1, 1, 4, 4, 3, 2, y)
  assert y.cpu().numpy()[0, 0, :, 0].tolist() == [5.0, 9.0]


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("nc,cin,with_probs", [(11, 32, False), (20, 64, False), (20, 64, True), (11, 64, True)])
def test_head(cuda, nc, cin, with_probs, math):
  _head_case(cuda, nc, cin, with_probs, math, 2, 9, 37)


@pytest.mark.first_hw_run
@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("nc,cin,with_probs,n,h,w", [(2, 16, True, 1, 1, 1), (5, 32, False, 1, 3, 130), (16, 64, True, 2, 4, 33),
                                                       (17, 32, False, 1, 7, 16), (33, 64, True, 1, 5, 48), (48, 32, False, 1, 2, 129),
                                                       (64, 128, True, 1, 3, 20), (3, 256, False, 1, 6, 17)])
def test_head_class_counts_and_shapes(cuda, nc, cin, with_probs, n, h, w, math):
  """One to four 16-class tiles (NUM_CLASS 2 .. 64), counts on both sides of every tile boundary, single-pixel and
  ragged images: the reference's configs use 4, 11, 20 and 34 classes."""
  _head_case(cuda, nc, cin, with_probs, math, n, h, w)


def _head_case(cuda, nc, cin, with_probs, math, n, h, w):
  import torch
  rng = np.random.default_rng(nc * 1000 + cin + with_probs)
  x = rnd(rng, n, h, w, cin)
  kern = (rng.standard_normal((3, 3, cin, nc)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
  bias = rng.normal(0, 0.05, nc).astype(np.float32)
  mask = rng.random((n, h, w)) < 0.8
  none_index = nc - 1
  logits64 = O.conv2d(x.astype(np.float64), kern.astype(np.float64), bias.astype(np.float64))
  prob64, pred64 = O.segmentation_head(logits64, mask, none_index)
  preds = torch.full((n, h, w), -1, dtype=torch.int32, device=cuda)
  logits = torch.empty((n, h, w, nc), dtype=torch.float32, device=cuda)
  probs = torch.empty((n, h, w, nc), dtype=torch.float32, device=cuda) if with_probs else None
  E.op_head(dev(x, cuda), dev(mask.astype(np.uint8), cuda), n, h, w, cin, kern, bias, none_index,
            preds, probs, logits, math)
  lg = logits.cpu().numpy()
  assert np.abs(lg - logits64).max() <= 2e-5 * max(1.0, np.abs(logits64).max())
  pr = preds.cpu().numpy()
  assert (pr[~mask] == none_index).all()
  srt = np.sort(logits64, -1)
  decided = (srt[..., -1] - srt[..., -2]) > 1e-4
  assert np.array_equal(pr[decided], pred64[decided])
  # on the device's own logits the argmax must be exact, first maximum winning
  own = np.where(mask, np.argmax(lg, -1), none_index)
  if not with_probs:
    assert np.array_equal(pr, own)
  else:
    assert np.abs(probs.cpu().numpy() - prob64).max() <= 1e-5


def test_head_tie_takes_lowest_index(cuda):
  """Two identical class columns: argmax must return the lower index (tf.argmax rule)."""
  import torch
  cin, nc = 16, 11
  rng = np.random.default_rng(3)
  x = rnd(rng, 1, 4, 16, cin)
  kern = (rng.standard_normal((3, 3, cin, nc)) * 0.1).astype(np.float32)
  bias = np.zeros(nc, np.float32)
  kern[..., 7] = kern[..., 2]
  kern[..., 2] *= 1.0
  bias[2] = bias[7] = 50.0            # classes 2 and 7 tie and dominate everywhere
  mask = np.ones((1, 4, 16), np.uint8)
  for with_probs in (False, True):
    preds = torch.full((1, 4, 16), -1, dtype=torch.int32, device=cuda)
    probs = torch.empty((1, 4, 16, nc), dtype=torch.float32, device=cuda) if with_probs else None
    E.op_head(dev(x, cuda), dev(mask, cuda), 1, 4, 16, cin, kern, bias, 10, preds, probs, None)
    assert (preds.cpu().numpy() == 2).all()
