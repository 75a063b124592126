"""SURVEY.md §8 row a1 pinned to the reference itself: tests/golden/preproc_*.npz hold the outputs of the
reference's own ``DataLoader.parse_sample`` (data_loader/data_loader.py:138-187, twin of inference.py:47-66) on
real scans of its dataset_samples/ (generator: tests/golden/make_preproc_golden.py, build container only).
CPU: the oracle restatement reproduces them bit for bit.  GPU (-m gpu): so do pclseg_op_normalize, the
``mask_out`` of pclseg_forward_raw and the device-side ``label[~mask] = None`` step of the eval loop."""
import glob
import os

import numpy as np
import pytest

from oracle import np_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[len("preproc_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "preproc_*.npz")))


def test_fixtures_present():
  assert CASES == ["ika_train_32x240", "ika_val_32x240", "kitti_val_64x1024", "nuscenes_val_32x1024"]


@pytest.mark.parametrize("case", CASES)
def test_oracle_preprocessing_equals_the_reference_bit_for_bit(case):
  g = np.load(os.path.join(GOLDEN, "preproc_%s.npz" % case))
  lidar, mask, label, weight = O.parse_sample(g["sample"], g["mean"], g["std"], int(g["none_index"]), g["cls_loss_weight"])
  assert lidar.dtype == np.float32 and np.array_equal(lidar.view(np.uint32), g["lidar"].view(np.uint32))   # incl. the sign of zero
  assert np.array_equal(mask, g["mask"]) and np.array_equal(label, g["label"]) and np.array_equal(weight, g["weight"])
  # the two-output form the network parity tests use is the same arithmetic
  l64, m = O.normalize_and_mask(g["sample"], g["mean"], g["std"])
  assert np.array_equal(l64.astype(np.float32), g["lidar"]) and np.array_equal(m, g["mask"])
  assert (g["label"][~g["mask"]] == int(g["none_index"])).all()
  if case == "ika_val_32x240":      # float64 file content that float32 cannot hold: the cast precedes the maths
    assert g["sample"].dtype == np.float64
    late = ((g["sample"][..., :5] - g["mean"]) / g["std"]).astype(np.float32)
    late[~g["mask"]] = 0
    assert not np.array_equal(late, g["lidar"][..., :5])


@pytest.mark.gpu
@pytest.mark.first_hw_run
@pytest.mark.parametrize("case", CASES)
def test_device_preprocessing_equals_the_reference_bit_for_bit(cuda, case):
  import torch
  import pclsegmentation_amd as P
  from pclsegmentation_amd import engine as E
  from pclsegmentation_amd.eval import masked_labels
  g = np.load(os.path.join(GOLDEN, "preproc_%s.npz" % case))
  sample = torch.from_numpy(g["sample"].astype(np.float32)).to(cuda)          # the reference's first step (:153)
  n, h, w, _ = sample.shape
  raw = sample[..., :5].contiguous()
  out = torch.empty((n, h, w, 6), dtype=torch.float32, device=cuda)
  mask = torch.empty((n, h, w), dtype=torch.uint8, device=cuda)
  E.op_normalize(raw, n, h, w, g["mean"], g["std"], out, mask)
  assert np.array_equal(out.cpu().numpy().view(np.uint32), g["lidar"].view(np.uint32))
  assert np.array_equal(mask.cpu().numpy().astype(bool), g["mask"])
  # the same mask leaves the network entry (pclseg_forward_raw's mask_out), device and host boundary
  mc, model = P.load_model_config("squeezesegv2", str(g["config"]), height=h, width=w)
  assert np.array_equal(np.asarray(mc.INPUT_MEAN, np.float64).ravel(), g["mean"])
  assert np.array_equal(np.asarray(mc.INPUT_STD, np.float64).ravel(), g["std"])
  model.init_weights(4321)
  preds, m2 = model.predict_raw(raw, return_mask=True)
  assert np.array_equal(m2.cpu().numpy().astype(bool), g["mask"])
  p_host, m_host = model.predict_raw(raw.cpu().numpy(), return_mask=True)
  assert np.array_equal(m_host, g["mask"]) and np.array_equal(np.asarray(p_host), preds.cpu().numpy())
  none_index = int(g["none_index"])
  assert (preds.cpu().numpy()[~g["mask"]] == none_index).all()                  # SegmentationNetwork.py:66-68
  # label[~mask] = None, as the eval loop does it on the device
  label = masked_labels(sample, m2, none_index)
  assert label.dtype == torch.int32 and np.array_equal(label.cpu().numpy(), g["label"])
  model._drop_engines()
