"""Whole-network parity on the MI355X against the committed golden vectors
(tests/golden/model_*.npz, made by tests/golden/make_model_golden.py from the float64
oracle) and, live, against the oracle including every intermediate activation.

Acceptance (BASELINE.json north_star / SURVEY.md §8(c)): max |logit - oracle| <= 1e-3 and
identical class IDs wherever the oracle's top-1/top-2 logit margin exceeds 2e-3.
"""
import os

import numpy as np
import pytest

import pclsegmentation_amd as P
from oracle import np_oracle as O
from pclsegmentation_amd import engine as E
from pclsegmentation_amd.utils.synthetic import synthetic_scans
from conftest import device_sync, host_tensor

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOGIT_TOL = 1e-3
MARGIN = 2e-3

CASES = [
  ("ssv2_32x240", "squeezesegv2", "squeezesegv2"),
  ("ssv2kitti_64x256", "squeezesegv2", "squeezesegv2kitti"),
  ("ssv2_real_32x240", "squeezesegv2", "squeezesegv2"),
  ("darknet21_32x240", "darknet21", "darknet21"),
  ("darknet53_32x240", "darknet53", "darknet53"),
  ("darknet53kitti_16x64", "darknet53", "darknet53kitti"),
]


def run_engine(model, raw, micro_batch=0, flags=0):
  """raw host scans -> (preds, logits, mask) through pclseg_forward_raw."""
  n, h, w, _ = raw.shape
  model.micro_batch = micro_batch
  eng = model.engine(h, w, flags)
  preds = np.empty((n, h, w), np.int32)
  logits = np.empty((n, h, w, model.NUM_CLASS), np.float32)
  mask = np.empty((n, h, w), np.uint8)
  eng.forward_raw(np.ascontiguousarray(raw), n, preds, None, logits, mask, mem=E.MEM_HOST)
  return preds, logits, mask.astype(bool), eng


def check_against(preds, logits, mask, gold_logits, gold_preds, gold_margin, none_index):
  err = np.abs(logits - gold_logits).max()
  assert err <= LOGIT_TOL, "max |logit - oracle| = %g" % err
  assert (preds[~mask] == none_index).all()
  decided = gold_margin > MARGIN
  assert np.array_equal(preds[decided], gold_preds[decided])
  return err


@pytest.mark.parametrize("flags", [0, E.FLAG_EXACT_F32], ids=["f16x3", "f32"])
@pytest.mark.parametrize("name,model_name,config_name", CASES, ids=[c[0] for c in CASES])
def test_golden(cuda, name, model_name, config_name, flags):
  g = np.load(os.path.join(GOLDEN, "model_%s.npz" % name))
  mc, model = P.load_model_config(model_name, config_name)
  model.init_weights(4321)
  preds, logits, mask, _ = run_engine(model, g["raw"], flags=flags)
  assert np.array_equal(mask, g["mask"])
  check_against(preds, logits, mask, g["logits"], g["preds"], g["margin"], mc.CLASSES.index("None"))


@pytest.mark.parametrize("flags", [0, E.FLAG_EXACT_F32], ids=["f16x3", "f32"])
@pytest.mark.parametrize("model_name,config_name,h,w", [
  ("squeezesegv2", "squeezesegv2", 32, 240), ("darknet21", "darknet21", 32, 64),
  # ragged against every tile shape: H % 4 != 0 (CAM / pool row blocks), CAM widths 104 = 4 x 26
  # and 52 = 2 x 26 exactly, conv tiles overhanging on both axes
  ("squeezesegv2", "squeezesegv2", 30, 208)])
def test_every_intermediate_matches_oracle(cuda, model_name, config_name, h, w, flags):
  """Layer-by-layer comparison (engine built with KEEP_ACTIVATIONS)."""
  mc, model = P.load_model_config(model_name, config_name, height=h, width=w)
  model.init_weights(4321)
  raw = synthetic_scans(2, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=99)
  preds, logits, mask, eng = run_engine(model, raw, flags=flags | E.FLAG_KEEP_ACTIVATIONS)
  lidar, omask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  taps = {}
  O.forward(model.arch_name(), model.weights, lidar, omask, mc.CLASSES.index("None"),
            num_layers=mc.get("NUM_LAYERS"), dtype=np.float64, taps=taps)
  names = [t[0] for t in eng.tensors()]
  report = []
  for tap, want in taps.items():
    if tap == "logits" or tap not in names:
      continue
    got = eng.read_tensor(names.index(tap))
    report.append((tap, float(np.abs(got - want).max()), float(np.abs(want).max())))
  bad = [r for r in report if r[1] > 1e-4 * max(1.0, r[2])]
  assert report and not bad, "first mismatching tensors: %s" % bad[:5]
  assert np.abs(logits - taps["logits"]).max() <= LOGIT_TOL


def test_micro_batch_invariance(cuda):
  """Any split of the batch into micro-batches gives bit-identical outputs."""
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  raw = synthetic_scans(7, 32, 240, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=5)
  ref = None
  for mb in (1, 2, 3, 5, 16):     # 7 scans: uneven splits, more / fewer micro-batches than lanes
    model._drop_engines()
    preds, logits, _, _ = run_engine(model, raw, micro_batch=mb)
    if ref is None:
      ref = (preds.copy(), logits.copy())
    else:
      assert np.array_equal(preds, ref[0]) and np.array_equal(logits, ref[1])


def test_model_call_surface(cuda):
  """probabilities, predictions = model([lidar, mask]) — host arrays and device tensors."""
  import torch
  if os.environ.get("HIPSIM_STREAMS") == "lazy":
    # the float32 / uint8 copies the model makes of its inputs are freed when the call returns; torch's caching
    # allocator on a real device orders their reuse after the stream's queued work, a CPU tensor's memory is recycled at
    # once and the simulator's deferred kernels would read it after that: a property of the harness, not of the engine
    pytest.skip("simulator, lazy streams: CPU tensors standing in for device tensors have no stream-ordered lifetime")
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  raw = synthetic_scans(2, 32, 240, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=11)
  lidar, mask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)   # float64, like inference.py
  probabilities, predictions = model([lidar, mask])
  assert predictions.numpy().dtype == np.int32 and predictions.shape == (2, 32, 240)
  assert probabilities.numpy().shape == (2, 32, 240, 11)
  assert np.allclose(probabilities.numpy().sum(-1), 1.0, atol=1e-5)
  assert np.array_equal(np.where(mask, probabilities.numpy().argmax(-1), 10), predictions.numpy())
  # raw entry point agrees with the reference-shaped one
  assert np.array_equal(model.predict_raw(raw).numpy(), predictions.numpy())
  # device tensors in -> device tensors out, same numbers
  pt, pr = model([torch.from_numpy(lidar).to(cuda), torch.from_numpy(mask).to(cuda)])
  device_sync(cuda)
  assert E.on_device(pr) and np.array_equal(pr.cpu().numpy(), predictions.numpy())
  assert np.array_equal(pt.cpu().numpy(), probabilities.numpy())
  with pytest.raises(ValueError):
    model([lidar[:, :, :200], mask[:, :, :200]])          # W % 16 != 0


def test_idempotent_and_deterministic(cuda):
  mc, model = P.load_model_config("darknet21", "darknet21", height=32, width=64)
  model.init_weights(4321)
  raw = synthetic_scans(3, 32, 64, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=2)
  a = run_engine(model, raw)
  b = run_engine(model, raw)
  assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_inference_cli_writes_reference_outputs(cuda, tmp_path):
  """The CLI keeps the reference's flags and per-scan outputs (reference: inference.py:81-112):
  pred_<name>.npy int32 [H,W], plot_<name>.png and plot_gt_<name>.png (RGBA)."""
  from PIL import Image
  from pclsegmentation_amd import inference as cli
  g = np.load(os.path.join(GOLDEN, "model_ssv2_real_32x240.npz"))
  src = tmp_path / "scans"
  src.mkdir()
  for i in range(2):   # real scans of the reference's sample dataset, with their label channel
    sample = np.concatenate([g["raw"][i], g["labels"][i][..., None].astype(np.float32)], -1)
    np.save(str(src / ("scan%d.npy" % i)), sample.astype(np.float64))
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  model.save(str(tmp_path / "model.npz"))
  out = tmp_path / "out"
  cli.main(["-d", str(src / "*.npy"), "-m", "squeezesegv2", "-t", str(out), "-p", str(tmp_path / "model.npz")])
  for i in range(2):
    pred = np.load(str(out / ("pred_scan%d.npy" % i)))
    assert pred.dtype == np.int32 and pred.shape == (32, 240)
    decided = g["margin"][i] > MARGIN
    assert np.array_equal(pred[decided], g["preds"][i][decided])
    for prefix in ("plot_", "plot_gt_"):
      img = Image.open(str(out / ("%sscan%d.png" % (prefix, i))))
      assert img.mode == "RGBA" and img.size == (240, 32)
    rgb = np.asarray(Image.open(str(out / ("plot_scan%d.png" % i))))[..., :3]
    assert np.array_equal(rgb, (255 * np.asarray(mc.CLS_COLOR_MAP)[pred]).astype(np.uint8))


def test_full_size_kitti_shape(cuda):
  """BASELINE configs[1] at its real size (SqueezeSegV2, 64x2048, 20 classes, batch 32):
  one scan against the float64 oracle, and size-independent properties over the whole batch —
  every scan's result is independent of the batch it travels in (micro-batch / lane placement),
  masked pixels carry the None class, and a repeat run is bit-identical."""
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2kitti", height=64, width=2048)
  model.init_weights(4321)
  raw = synthetic_scans(32, 64, 2048, mc.INPUT_MEAN, mc.INPUT_STD, 0.78, seed=1234)
  preds, logits, mask, _ = run_engine(model, raw)
  assert (preds[~mask] == 0).all() and preds.min() >= 0 and preds.max() < 20
  # (a) the float64 oracle on five scans, one per position of a micro-batch and spread over the lanes
  # (micro-batches of 4 scans dealt round-robin to 3 lanes: scan -> (micro-batch, position) 0 -> (0,0), 7 -> (1,3),
  # 14 -> (3,2), 21 -> (5,1), 31 -> (7,3)); the rest of the batch is covered by (b) and (c)
  for i in (0, 7, 14, 21, 31):
    lidar, omask = O.normalize_and_mask(raw[i:i + 1], mc.INPUT_MEAN, mc.INPUT_STD)
    _, opred, ologits = O.forward("squeezesegv2", model.weights, lidar, omask, 0, dtype=np.float64)
    assert np.array_equal(omask[0], mask[i])
    srt = np.sort(ologits[0], -1)
    check_against(preds[i], logits[i], mask[i], ologits[0], opred[0], (srt[..., -1] - srt[..., -2]), 0)
  # (b) batch-composition invariance: scans 7, 19, 31 alone == inside the batch of 32
  for i in (7, 19, 31):
    p1, l1, _, _ = run_engine(model, raw[i:i + 1])
    assert np.array_equal(p1[0], preds[i]) and np.array_equal(l1[0], logits[i])
  # (c) determinism of the multi-lane schedule
  p2, l2, _, _ = run_engine(model, raw)
  assert np.array_equal(p2, preds) and np.array_equal(l2, logits)


def test_c1_reference_sample_dataset_all_32_scans(cuda):
  """BASELINE configs[0]: SqueezeSegV2 / 11 classes on ALL 32 real scans of the reference's
  dataset_samples/sample_dataset/train (32x240; fixture = the inputs, made by make_c1_fixture.py), as
  one batch of 32 and in the reference's own loop shape — one scan per call (inference.py:44-75) —
  against the float64 oracle, both arithmetic modes."""
  g = np.load(os.path.join(GOLDEN, "c1_sample_dataset_train_32x240.npz"))
  raw = g["raw"]
  assert raw.shape == (32, 32, 240, 5)
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  none_index = mc.CLASSES.index("None")
  lidar, omask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  _, opred, ologits = O.forward("squeezesegv2", model.weights, lidar, omask, none_index, dtype=np.float64)
  srt = np.sort(ologits, -1)
  margin = srt[..., -1] - srt[..., -2]
  for flags in (0, E.FLAG_EXACT_F32):
    preds, logits, mask, _ = run_engine(model, raw, flags=flags)
    assert np.array_equal(mask, omask)
    check_against(preds, logits, mask, ologits, opred, margin, none_index)
    for i in (0, 13, 31):                                   # the reference's batch-1 loop
      p1, l1, _, _ = run_engine(model, raw[i:i + 1], flags=flags)
      assert np.array_equal(p1[0], preds[i]) and np.array_equal(l1[0], logits[i])
    model._drop_engines()


def test_full_size_darknet21_nuscenes_shape(cuda):
  """BASELINE configs[4]: Darknet-21 at 32x1024 — one scan against the oracle plus batch
  invariance."""
  mc, model = P.load_model_config("darknet21", "darknet21", height=32, width=1024)
  model.init_weights(4321)
  raw = synthetic_scans(64, 32, 1024, mc.INPUT_MEAN, mc.INPUT_STD, 0.59, seed=1234)   # the config's batch 64
  preds, logits, mask, _ = run_engine(model, raw)
  assert (preds[~mask] == 10).all() and preds.min() >= 0 and preds.max() < 11
  for i in (2, 29, 63):          # three scans against the float64 oracle (different micro-batches and lanes)
    lidar, omask = O.normalize_and_mask(raw[i:i + 1], mc.INPUT_MEAN, mc.INPUT_STD)
    _, opred, ologits = O.forward("darknet21", model.weights, lidar, omask, 10, num_layers=21, dtype=np.float64)
    srt = np.sort(ologits[0], -1)
    check_against(preds[i], logits[i], mask[i], ologits[0], opred[0], (srt[..., -1] - srt[..., -2]), 10)
  for i in (2, 40, 63):
    p1, l1, _, _ = run_engine(model, raw[i:i + 1])
    assert np.array_equal(p1[0], preds[i]) and np.array_equal(l1[0], logits[i])


def test_inference_cli_reads_a_savedmodel_directory(cuda, tmp_path):
  """-p may be a reference SavedModel directory (inference.py:39): the tensor bundle is read
  without TensorFlow (fixture from tests/bundle_writer.py — reader parity is unpinned)."""
  from pclsegmentation_amd import inference as cli
  from tests.test_savedmodel import _make_savedmodel
  g = np.load(os.path.join(GOLDEN, "model_ssv2_real_32x240.npz"))
  src = tmp_path / "scans"
  src.mkdir()
  np.save(str(src / "scan0.npy"), g["raw"][0].astype(np.float64))
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  _make_savedmodel(str(tmp_path / "saved"), model.weight_spec(), model.weights)
  out = tmp_path / "out"
  cli.main(["-d", str(src / "*.npy"), "-m", "squeezesegv2", "-t", str(out), "-p", str(tmp_path / "saved"),
            "--no_plots"])
  pred = np.load(str(out / "pred_scan0.npy"))
  decided = g["margin"][0] > MARGIN
  assert np.array_equal(pred[decided], g["preds"][0][decided])


def test_full_size_darknet53_kitti_shape(cuda):
  """BASELINE configs[2] at its real size (Darknet-53, 64x2048, 20 classes, batch 16): one scan
  against the float64 oracle, batch-composition invariance and a bit-identical repeat.  This is the
  size at which the deep 64x128-pixel layers run group-major over a 212 MB weight set and the
  batch is split into micro-batches across the lanes."""
  mc, model = P.load_model_config("darknet53", "darknet53kitti", height=64, width=2048)
  model.init_weights(4321)
  raw = synthetic_scans(16, 64, 2048, mc.INPUT_MEAN, mc.INPUT_STD, 0.78, seed=1234)
  preds, logits, mask, _ = run_engine(model, raw)
  none_index = mc.CLASSES.index("None")
  assert (preds[~mask] == none_index).all() and preds.min() >= 0 and preds.max() < mc.NUM_CLASS
  for i in (5, 12):             # two scans against the float64 oracle (22 s of NumPy each), different lanes
    lidar, omask = O.normalize_and_mask(raw[i:i + 1], mc.INPUT_MEAN, mc.INPUT_STD)
    _, opred, ologits = O.forward("darknet53", model.weights, lidar, omask, none_index, num_layers=53,
                                  dtype=np.float64)
    srt = np.sort(ologits[0], -1)
    check_against(preds[i], logits[i], mask[i], ologits[0], opred[0], (srt[..., -1] - srt[..., -2]), none_index)
  for i in (5, 15):
    p1, l1, _, _ = run_engine(model, raw[i:i + 1])
    assert np.array_equal(p1[0], preds[i]) and np.array_equal(l1[0], logits[i])
  p2, l2, _, _ = run_engine(model, raw)
  assert np.array_equal(p2, preds) and np.array_equal(l2, logits)


@pytest.mark.parametrize("stride", [8, 32])
def test_darknet_output_strides(cuda, stride):
  """OUTPUT_STRIDE 8 and 32 (nets/Darknet.py:158-181,215-231: which encoder layers shrink W and
  which decoder layers grow it) against the oracle on every intermediate tensor."""
  mc, model = P.load_model_config("darknet21", "darknet21", height=16, width=128)
  mc.OUTPUT_STRIDE = stride
  model.init_weights(4321)
  raw = synthetic_scans(2, 16, 128, mc.INPUT_MEAN, mc.INPUT_STD, 0.7, seed=21)
  preds, logits, mask, eng = run_engine(model, raw, flags=E.FLAG_KEEP_ACTIVATIONS)
  lidar, omask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  taps = {}
  _, opred, ologits = O.forward("darknet21", model.weights, lidar, omask, mc.CLASSES.index("None"),
                                num_layers=21, output_stride=stride, dtype=np.float64, taps=taps)
  names = [t[0] for t in eng.tensors()]
  checked = 0
  for tap, want in taps.items():
    if tap == "logits" or tap not in names:
      continue
    got = eng.read_tensor(names.index(tap))
    assert got.shape == want.shape, (tap, got.shape, want.shape)
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), tap
    checked += 1
  assert checked > 20
  assert np.abs(logits - ologits).max() <= LOGIT_TOL


def test_nan_pixel_never_yields_an_out_of_range_class(cuda):
  """A scan pixel with depth > 0 but NaN coordinates (a corrupt file) makes the logits of every pixel
  in its receptive field NaN.  The reference's softmax then yields NaN probabilities and tf.argmax
  over them index 0; the predictions must stay inside [0, NC) (here: 0 where valid) and every other
  pixel must keep its regular result."""
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  raw = synthetic_scans(2, 32, 240, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=8)
  clean_preds, clean_logits, _, _ = run_engine(model, raw)
  bad = raw.copy()
  bad[1, 10, 100, :3] = np.nan
  bad[1, 10, 100, 4] = 7.0
  for flags in (0, E.FLAG_EXACT_F32):
    model._drop_engines()
    preds, logits, mask, _ = run_engine(model, bad, flags=flags)
    assert preds.min() >= 0 and preds.max() < mc.NUM_CLASS
    nan_px = np.isnan(logits).any(-1)
    assert nan_px[1].any() and not nan_px[0].any()
    assert (preds[nan_px & mask] == 0).all() and (preds[~mask] == mc.CLASSES.index("None")).all()
    if flags == 0:
      assert np.array_equal(preds[0], clean_preds[0]) and np.array_equal(logits[0], clean_logits[0])
    probs, p2 = model([*O.normalize_and_mask(bad, mc.INPUT_MEAN, mc.INPUT_STD)])
    assert np.isnan(probs.numpy()[nan_px]).all() and np.array_equal(p2.numpy(), preds)
  model._drop_engines()


def test_split_f16_range_guard_and_exact_fallback(cuda):
  """Default arithmetic carries operands as f16 hi/lo pairs and needs |activation| < 65504.  With
  conv1's kernel scaled so that activations leave that range the default engine must SAY so
  (PCLSEG_ERR_RANGE -> FloatingPointError) instead of returning NaN-derived classes; exact-f32 mode
  still matches the oracle, and PCLSEG_FLAG_RANGE_FALLBACK repairs the call transparently."""
  import torch
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  w = dict(model.weights)
  w["conv1/kernel"] = w["conv1/kernel"] * np.float32(3.0e4)
  model.set_weights(w)
  raw = synthetic_scans(2, 32, 240, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=13)
  lidar, omask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  taps = {}
  _, opred, ologits = O.forward("squeezesegv2", model.weights, lidar, omask, 10, dtype=np.float64, taps=taps)
  assert np.abs(taps["conv1"]).max() > 65504          # the premise: out of the f16 range
  scale = np.abs(ologits).max()
  srt = np.sort(ologits, -1)
  decided = (srt[..., -1] - srt[..., -2]) > 1e-4 * scale
  with pytest.raises(FloatingPointError):
    run_engine(model, raw)
  model._drop_engines()
  p_exact, l_exact, _, _ = run_engine(model, raw, flags=E.FLAG_EXACT_F32)
  assert np.abs(l_exact - ologits).max() <= 2e-5 * scale
  assert np.array_equal(p_exact[decided], opred[decided])
  model._drop_engines()
  p_fb, l_fb, _, _ = run_engine(model, raw, flags=E.FLAG_RANGE_FALLBACK)
  assert np.array_equal(p_fb, p_exact) and np.array_equal(l_fb, l_exact)
  # asynchronous (device-memory) calls: the guard is reported / repaired by sync()
  for flags, expect_error in ((0, True), (E.FLAG_RANGE_FALLBACK, False)):
    model._drop_engines()
    eng = model.engine(32, 240, flags)
    d_raw = torch.from_numpy(raw).to(cuda)
    d_preds = torch.empty((2, 32, 240), dtype=torch.int32, device=cuda)
    eng.set_stream(E.stream_handle(cuda))
    eng.forward_raw(d_raw, 2, d_preds, None, None, None, mem=E.MEM_DEVICE)
    if expect_error:
      with pytest.raises(FloatingPointError):
        eng.sync()
      eng.sync()                                        # the flag is sticky until reported, then cleared
    else:
      eng.sync()
      assert np.array_equal(d_preds.cpu().numpy(), p_exact)
  model._drop_engines()


def test_range_fallback_repairs_every_unsynced_call(cuda):
  """the range flag is one sticky word shared by all queued calls.  Two device calls are
  enqueued before one sync — the FIRST overflows the f16 range, the second (an all-invalid scan: zero
  input, activations = biases) does not.  A fallback handle must repair the first call too, not only
  the last one; so must a synchronous MEM_HOST call that observes a flag raised by a pending device call."""
  import torch
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  w = dict(model.weights)
  w["conv1/kernel"] = w["conv1/kernel"] * np.float32(3.0e4)
  model.set_weights(w)
  hot = synthetic_scans(2, 32, 240, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=13)
  cold = hot.copy()
  cold[..., 4] = 0.0                                      # depth 0 everywhere: every pixel masked, lidar = 0
  p_hot, _, _, _ = run_engine(model, hot, flags=E.FLAG_EXACT_F32)
  p_cold, _, _, _ = run_engine(model, cold, flags=E.FLAG_EXACT_F32)
  model._drop_engines()
  run_engine(model, cold)                                 # the premise: the cold scan alone stays in range
  model._drop_engines()
  eng = model.engine(32, 240, E.FLAG_RANGE_FALLBACK)
  eng.set_stream(E.stream_handle(cuda))
  d_hot, d_cold = torch.from_numpy(hot).to(cuda), torch.from_numpy(cold).to(cuda)
  o1 = torch.full((2, 32, 240), -7, dtype=torch.int32, device=cuda)
  o2 = torch.full((2, 32, 240), -7, dtype=torch.int32, device=cuda)
  eng.forward_raw(d_hot, 2, o1, None, None, None, mem=E.MEM_DEVICE)    # overflows
  eng.forward_raw(d_cold, 2, o2, None, None, None, mem=E.MEM_DEVICE)   # does not
  eng.sync()
  assert np.array_equal(o1.cpu().numpy(), p_hot), "the earlier of two un-synchronised calls was not repaired"
  assert np.array_equal(o2.cpu().numpy(), p_cold)
  # a synchronous host call behind a pending device call consumes the flag: both are repaired
  o1.fill_(-7)
  h_out = np.empty((2, 32, 240), np.int32)
  eng.forward_raw(d_hot, 2, o1, None, None, None, mem=E.MEM_DEVICE)
  eng.forward_raw(np.ascontiguousarray(cold), 2, h_out, None, None, None, mem=E.MEM_HOST)
  assert np.array_equal(h_out, p_cold)
  eng.sync()
  assert np.array_equal(o1.cpu().numpy(), p_hot)
  # without the fallback the synchronous call reports the flag and says earlier calls are suspect
  model._drop_engines()
  eng = model.engine(32, 240, 0)
  eng.set_stream(E.stream_handle(cuda))
  eng.forward_raw(d_hot, 2, o1, None, None, None, mem=E.MEM_DEVICE)
  with pytest.raises(FloatingPointError, match="asynchronous"):
    eng.forward_raw(np.ascontiguousarray(cold), 2, h_out, None, None, None, mem=E.MEM_HOST)
  eng.sync()
  model._drop_engines()


def test_finalize_rejects_weights_the_split_cannot_carry(cuda):
  """a BatchNorm-folded weight that is not finite (moving_variance = -eps makes
  gamma / sqrt(var + eps) infinite) cannot be split into f16 hi/lo (inf - inf = NaN): finalize says so
  instead of shipping NaN fragments; a RANGE_FALLBACK handle runs such a model in exact float32."""
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  w = dict(model.weights)
  v = w["fire2/expand3x3_bn/moving_variance"].copy()
  v[3] = np.float32(-1e-3)
  w["fire2/expand3x3_bn/moving_variance"] = v
  model.set_weights(w)
  raw = synthetic_scans(1, 32, 240, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=5)
  with pytest.raises(FloatingPointError, match="not finite"):
    run_engine(model, raw)
  model._drop_engines()
  p_exact, l_exact, _, _ = run_engine(model, raw, flags=E.FLAG_EXACT_F32)
  model._drop_engines()
  p_fb, l_fb, _, _ = run_engine(model, raw, flags=E.FLAG_RANGE_FALLBACK)
  assert np.array_equal(p_fb, p_exact) and np.array_equal(l_fb, l_exact, equal_nan=True)
  model._drop_engines()
  # huge but finite folded weights are fine: the per-channel scale brings them into range
  w = dict(model.weights)
  w["fire2/expand3x3_bn/moving_variance"] = model.weights["fire2/expand3x3_bn/moving_variance"] * 0 + np.float32(1.0)
  g = w["fire2/expand3x3_bn/gamma"].copy(); g[3] = np.float32(3.0e6)
  w["fire2/expand3x3_bn/gamma"] = g
  model.set_weights(w)
  run_engine(model, raw, flags=E.FLAG_RANGE_FALLBACK)     # finalize accepts (whether the ACTIVATIONS overflow is the range guard's business)
  model._drop_engines()


def test_host_boundary_pinned_and_pageable_agree_with_device(cuda):
  """PCLSEG_MEM_HOST with page-locked buffers (DMA uploads where a slot is free, copy kernels and direct
  prediction writes over PCIe where a DMA command would block), with pageable buffers (pinned bounce
  slabs) and PCLSEG_MEM_DEVICE give bit-identical outputs, for a batch that spans several micro-batches
  per lane and has a ragged tail."""
  import torch
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.init_weights(4321)
  n, h, w = 11, 32, 240
  raw = synthetic_scans(n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=17)
  model.micro_batch = 2
  eng = model.engine(h, w)
  d_raw = torch.from_numpy(raw).to(cuda)
  d_preds = torch.empty((n, h, w), dtype=torch.int32, device=cuda)
  d_logits = torch.empty((n, h, w, mc.NUM_CLASS), dtype=torch.float32, device=cuda)
  eng.set_stream(E.stream_handle(cuda))
  eng.forward_raw(d_raw, n, d_preds, None, d_logits, None, mem=E.MEM_DEVICE)
  eng.sync()
  want_p, want_l = d_preds.cpu().numpy(), d_logits.cpu().numpy()
  lidar, omask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  for pinned in (True, False):
    mk = lambda *s, dtype: host_tensor(s, dtype, pinned)
    h_raw = mk(n, h, w, 5, dtype=torch.float32)
    h_raw.copy_(torch.from_numpy(raw))
    h_preds, h_logits = mk(n, h, w, dtype=torch.int32), mk(n, h, w, mc.NUM_CLASS, dtype=torch.float32)
    h_probs, h_mask = mk(n, h, w, mc.NUM_CLASS, dtype=torch.float32), mk(n, h, w, dtype=torch.uint8)
    for _ in range(2):    # second call reuses the staging slabs
      h_preds.zero_()
      eng.forward_raw(h_raw, n, h_preds, h_probs, h_logits, h_mask, mem=E.MEM_HOST)
      assert np.array_equal(h_logits.numpy(), want_l)
      assert np.array_equal(h_mask.numpy().astype(bool), omask)
      assert np.allclose(h_probs.numpy().sum(-1), 1.0, atol=1e-5)
      # with probabilities materialised the argmax runs over them (same classes off exact ties)
      srt = np.sort(want_l, -1)
      decided = (srt[..., -1] - srt[..., -2]) > 1e-6
      assert np.array_equal(h_preds.numpy()[decided], want_p[decided])
    if pinned:   # enqueue-only calls on page-locked buffers, two in a row without a wait in between
      h_preds2, h_logits2 = mk(n, h, w, dtype=torch.int32), mk(n, h, w, mc.NUM_CLASS, dtype=torch.float32)
      h_preds.zero_()
      eng.forward_raw(h_raw, n, h_preds, None, h_logits, None, mem=E.MEM_HOST_ASYNC)
      eng.forward_raw(h_raw, n, h_preds2, None, h_logits2, None, mem=E.MEM_HOST_ASYNC)
      eng.sync()
      for pp, ll in ((h_preds, h_logits), (h_preds2, h_logits2)):
        assert np.array_equal(pp.numpy(), want_p) and np.array_equal(ll.numpy(), want_l)
    else:
      with pytest.raises(ValueError):
        eng.forward_raw(h_raw, n, h_preds, None, None, None, mem=E.MEM_HOST_ASYNC)   # needs pinned buffers
    # the reference-shaped entry (normalised lidar + mask) through the same boundary
    h_lidar = mk(n, h, w, 6, dtype=torch.float32)
    h_lidar.copy_(torch.from_numpy(lidar.astype(np.float32)))
    h_m = mk(n, h, w, dtype=torch.uint8)
    h_m.copy_(torch.from_numpy(omask.astype(np.uint8)))
    eng.forward(h_lidar, h_m, n, h_preds, None, h_logits, mem=E.MEM_HOST)
    assert np.array_equal(h_preds.numpy(), want_p) and np.array_equal(h_logits.numpy(), want_l)
    if pinned:
      # predictions only, enqueue-only: no DMA command at all — a copy kernel on each lane reads the scans
      # (and, for the reference-shaped entry, the mask) over PCIe and the head writes the class IDs straight
      # into the caller's buffer; three calls back to back reuse every slot
      h_preds.zero_(); h_preds2.zero_()
      h_preds3 = mk(n, h, w, dtype=torch.int32)
      eng.forward_raw(h_raw, n, h_preds, None, None, None, mem=E.MEM_HOST_ASYNC)
      eng.forward(h_lidar, h_m, n, h_preds2, None, None, mem=E.MEM_HOST_ASYNC)
      eng.forward_raw(h_raw, n, h_preds3, None, None, None, mem=E.MEM_HOST_ASYNC)
      eng.sync()
      for pp in (h_preds, h_preds2, h_preds3):
        assert np.array_equal(pp.numpy(), want_p)
  model.micro_batch = 0
  model._drop_engines()


@pytest.mark.parametrize("h,w", [(32, 240), (30, 208), (64, 512)])
def test_fused_squeeze_intermediates(cuda, h, w):
  """fire4/6/7/8/9's and (after their skip add) fire10/11/12's expand blocks also compute the NEXT
  module's squeeze and never write their own output (conv_kernel FSQ); pool1/3/5 are taken while
  fire2/4/6's squeeze loads its input (pool_squeeze_kernel); cam2's block computes fire3's squeeze;
  the FIREUP pairs up-convolve their own input patch (conv_kernel UP).  KEEP_ACTIVATIONS normally switches that fusion off, so this runs a
  worker with the debug switch PCLSEG_FUSE_KEEP=1 and compares every fused squeeze output and the
  logits with the float64 oracle, on shapes with ragged 64-pixel tiles on both axes."""
  import subprocess
  import sys
  env = dict(os.environ, PCLSEG_FUSE_KEEP="1")
  r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fused_worker.py"),
                      str(h), str(w)], env=env, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
  assert r.stdout.count("/squeeze") == 12, r.stdout


@pytest.mark.parametrize("config_name", ["squeezesegv2", "squeezesegv2kitti"])   # 11 classes: one head tile; 20: two
@pytest.mark.parametrize("h,w", [(5, 16), (9, 48), (64, 16), (1, 32), (17, 80)])
def test_fully_fused_plan_on_tiny_and_ragged_shapes(cuda, config_name, h, w):
  """The default (fully fused, 19-launch) plan on shapes where every tile of every fused kernel overhangs the
  image: a single 16-pixel tile column, fewer rows than a tile, one row — logits, probabilities, class IDs and
  the None class of masked pixels against the float64 oracle."""
  mc, model = P.load_model_config("squeezesegv2", config_name, height=h, width=w)
  model.init_weights(4321)
  raw = synthetic_scans(3, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=h * 100 + w)
  eng = model.engine(h, w)
  assert E.plan(eng.desc)["num_ops"] == 19
  preds = np.empty((3, h, w), np.int32)
  logits = np.empty((3, h, w, mc.NUM_CLASS), np.float32)
  probs = np.empty_like(logits)
  eng.forward_raw(raw, 3, preds, probs, logits, None, mem=E.MEM_HOST)
  lidar, mask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  none_index = mc.CLASSES.index("None")
  _, opred, ologits = O.forward("squeezesegv2", model.weights, lidar, mask, none_index, dtype=np.float64)
  srt = np.sort(ologits, -1)
  check_against(preds, logits, mask, ologits, opred, srt[..., -1] - srt[..., -2], none_index)
  assert np.allclose(probs.sum(-1), 1.0, atol=1e-5)
  model._drop_engines()


def test_fusion_is_active_by_default_and_off_for_debug_reads(cuda):
  """The default SqueezeSegV2 plan launches 20 kernels per micro-batch (37 without the nine fused
  squeezes, the three fused pools, the four fused up-convolutions and the fused fire13 + conv14 + head tail);
  KEEP_ACTIVATIONS / exact-f32 / range-fallback plans keep the 37-launch graph."""
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  d = E.make_desc("squeezesegv2", 32, 240, 11, 10, mc.INPUT_MEAN, mc.INPUT_STD)
  assert E.plan(d)["num_ops"] == 19           # + the normalise launch = 20
  for flags in (E.FLAG_KEEP_ACTIVATIONS, E.FLAG_EXACT_F32, E.FLAG_RANGE_FALLBACK):
    d = E.make_desc("squeezesegv2", 32, 240, 11, 10, mc.INPUT_MEAN, mc.INPUT_STD, flags=flags)
    assert E.plan(d)["num_ops"] == 36
