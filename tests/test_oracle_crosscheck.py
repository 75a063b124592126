"""The NumPy oracle against (a) the independent PyTorch-CPU expression of the same graph and
(b) the committed golden vectors.  Runs on CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle.torch_ref import TorchNet
from pclsegmentation_amd import configs as C
from pclsegmentation_amd.nets.weights import synthetic_weights, keras_default_weights, spec_for_config
from pclsegmentation_amd.utils.synthetic import synthetic_scans

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = [("squeezesegv2", C.SqueezeSegV2Config, 32, 240), ("squeezesegv2", C.SqueezeSegV2KittiConfig, 8, 64),
         ("darknet21", C.Darknet21, 8, 48), ("darknet53", C.Darknet53Kitti, 8, 32)]


@pytest.mark.parametrize("model,cfg,h,w", CASES, ids=[c[0] + "_%dx%d" % c[2:] for c in CASES])
def test_numpy_vs_torch_float64(model, cfg, h, w):
  mc = cfg()
  w_ = synthetic_weights(spec_for_config(model, mc))
  raw = synthetic_scans(2, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=21)
  lidar, mask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  _, pred, logits = O.forward(model, w_, lidar, mask, mc.CLASSES.index("None"),
                              num_layers=mc.get("NUM_LAYERS"), dtype=np.float64)
  net = TorchNet(model, w_, num_layers=mc.get("NUM_LAYERS"), dtype=torch.float64)
  assert np.abs(net.logits(lidar).numpy() - logits).max() <= 1e-10
  _, tpred = net(lidar, mask, mc.CLASSES.index("None"))
  assert np.array_equal(tpred, pred)


def test_numpy_float32_mode_close_to_float64():
  mc = C.SqueezeSegV2Config()
  w_ = synthetic_weights(spec_for_config("squeezesegv2", mc))
  raw = synthetic_scans(1, 32, 240, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=4)
  lidar, mask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  l64 = O.squeezesegv2_logits(w_, lidar, np.float64)
  l32 = O.squeezesegv2_logits(w_, lidar, np.float32)
  assert l32.dtype == np.float32 and np.abs(l32 - l64).max() <= 1e-4


def test_fresh_keras_initialisation_also_agrees():
  """gamma=1, beta=0, mean=0, var=1, glorot kernels, zero biases (what Model(mc) holds)."""
  mc = C.Darknet21()
  w_ = keras_default_weights(spec_for_config("darknet21", mc))
  raw = synthetic_scans(1, 8, 32, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=8)
  lidar, mask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  a = O.darknet_logits(w_, lidar, 21)
  b = TorchNet("darknet21", w_, dtype=torch.float64).logits(lidar).numpy()
  assert np.abs(a - b).max() <= 1e-10


@pytest.mark.parametrize("name,model,cfg", [
  ("ssv2_32x240", "squeezesegv2", C.SqueezeSegV2Config),
  ("ssv2_real_32x240", "squeezesegv2", C.SqueezeSegV2Config),
  ("darknet53kitti_16x64", "darknet53", C.Darknet53Kitti)])
def test_oracle_reproduces_golden_vectors(name, model, cfg):
  g = np.load(os.path.join(GOLDEN, "model_%s.npz" % name))
  mc = cfg()
  w_ = synthetic_weights(spec_for_config(model, mc))
  lidar, mask = O.normalize_and_mask(g["raw"], mc.INPUT_MEAN, mc.INPUT_STD)
  _, pred, logits = O.forward(model, w_, lidar, mask, mc.CLASSES.index("None"),
                              num_layers=mc.get("NUM_LAYERS"), dtype=np.float64)
  assert np.array_equal(mask, g["mask"])
  assert np.array_equal(logits.astype(np.float32), g["logits"])
  assert np.array_equal(pred, g["preds"])


def test_darknet_output_stride_variants():
  assert O.darknet_strides(16) == ([2, 2, 2, 2, 1], [1, 2, 2, 2, 2])
  assert O.darknet_strides(32) == ([2, 2, 2, 2, 2], [2, 2, 2, 2, 2])
  assert O.darknet_strides(8) == ([2, 2, 2, 1, 1], [1, 1, 2, 2, 2])


@pytest.mark.parametrize("name", ["kitti_64x1024", "small_32x256", "nuscenes_like_32x1024"])
def test_projection_oracle_reproduces_reference_outputs(name):
  """The projection restatement against outputs of the REFERENCE's own NumPy code
  (dataset_convert/laserscan_semantic_kitti.py, vectors made by make_projection_golden.py)."""
  g = np.load(os.path.join(GOLDEN, "projection_%s.npz" % name))
  rng_, xyz, rem, idx = O.range_projection(g["points"], int(g["H"]), int(g["W"]), float(g["fov_up"]),
                                           float(g["fov_down"]))
  assert np.array_equal(idx, g["proj_idx"])
  assert np.array_equal(rng_, g["proj_range"]) and np.array_equal(xyz, g["proj_xyz"])
  assert np.array_equal(rem, g["proj_remission"])


@pytest.mark.parametrize("name", ["ring_32x1024", "ring_16x256"])
def test_ring_projection_oracle_reproduces_the_reference(name):
  """tests/golden/projection2_ring_*.npz are outputs of the reference's own
  LaserScan.do_range_projection_ring / SemLaserScan.do_label_projection."""
  g = np.load(os.path.join(GOLDEN, "projection2_%s.npz" % name))
  rng_, xyz, rem, idx, mask = O.range_projection_ring(g["points"], g["ring"], int(g["H"]), int(g["W"]))
  assert np.array_equal(idx, g["proj_idx"]) and np.array_equal(rng_, g["proj_range"])
  assert np.array_equal(xyz, g["proj_xyz"]) and np.array_equal(rem, g["proj_remission"])
  assert np.array_equal(mask, g["proj_mask"])
  assert np.array_equal(O.label_projection(idx, g["labels"]), g["proj_sem_label"])


def test_label_projection_and_information_map_oracles_reproduce_the_reference():
  g = np.load(os.path.join(GOLDEN, "projection2_kitti_labels_32x512.npz"))
  lm = {int(k): int(v) for k, v in zip(g["map_keys"], g["map_values"])}
  assert np.array_equal(O.label_projection(g["proj_idx"], g["labels"]), g["proj_sem_label"])
  assert np.array_equal(O.label_projection(g["proj_idx"], g["labels"], lm), g["final"][..., 5])
  f = np.load(os.path.join(GOLDEN, "projection2_front_32x240.npz"))
  assert np.array_equal(O.information_map(f["pcl"]), f["info"])
