"""Evaluation on the engine (reference: eval.py:33-82) — mIoU / IoU / recall / precision over a
split of projected scans.  Each batch crosses PCIe ONCE (scans + labels up); the forward pass, the
``label[~mask] = None`` step and the confusion-matrix accumulation (``pclseg_op_confusion_matrix``,
exact int64 counts) all stay on the device, only the final [NC,NC] matrix comes back.

  python -m pclsegmentation_amd.eval -d <dataset dir> -i val -m squeezesegv2 -n squeezesegv2 -p model.npz

Same flags as the reference (``-d/-i/-t/-p/-m/-n``).  The reference reads a TFRecord file that
its DataLoader writes from ``<data_path>/<image_set>/*.npy``; here the ``.npy`` scans
([H,W,6]: x,y,z,intensity,depth,label) are read directly — the TFRecord pipeline is training
infrastructure and out of scope.  Like the reference (data_loader/data_loader.py:173-180) labels
of invalid pixels are set to the "None" class before counting.
"""
import argparse
import glob
import os

import numpy as np

from . import engine as _engine
from . import load_model, load_model_config


def confusion_matrix_to_iou_recall_precision(cm):
  """Classwise IoU, recall, precision from total_cm (rows = labels, columns = predictions).
  reference: utils/util.py:64-79 (tf.math.divide_no_nan semantics)."""
  cm = np.asarray(cm, np.float64)
  sum_over_col = cm.sum(axis=1)
  sum_over_row = cm.sum(axis=0)
  tp = np.diag(cm)
  fp = sum_over_row - tp
  fn = sum_over_col - tp

  def dnn(a, b):
    return np.where(b != 0, a / np.where(b != 0, b, 1), 0.0)

  return dnn(tp, tp + fp + fn), dnn(tp, tp + fn), dnn(tp, tp + fp)


class MeanIoU:
  """Device-side counterpart of ``tf.metrics.MeanIoU`` as the reference uses it
  (eval.py:41,48,50,58; nets/SegmentationNetwork.py:52): ``update_state(label, predictions)``
  accumulates ``total_cm``; ``result()`` is the mean IoU over classes that occur."""

  def __init__(self, num_classes, device=0, name="MeanIoU"):
    import torch
    self.num_classes = int(num_classes)
    self.name = name
    self._dev = _engine.torch_device(device)
    self._cm = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64, device=self._dev)
    self._seen = 0

  def reset_state(self):
    self._cm.zero_()
    self._seen = 0

  def update_state(self, label, predictions):
    """Host arrays or device tensors.  Labels outside [0, num_classes) are NOT counted (tf.metrics.
    MeanIoU would raise for them): ``ignored`` keeps their running total so a caller can tell."""
    import torch
    lab = torch.as_tensor(np.asarray(label) if not hasattr(label, "is_cuda") else label)
    prd = torch.as_tensor(np.asarray(predictions) if not hasattr(predictions, "is_cuda") else predictions)
    lab = lab.to(self._dev, dtype=torch.int32).contiguous()
    prd = prd.to(self._dev, dtype=torch.int32).contiguous()
    if lab.numel() != prd.numel():
      raise ValueError("label and predictions differ in size: %d vs %d" % (lab.numel(), prd.numel()))
    self._seen += lab.numel()
    _engine.op_confusion_matrix(lab, prd, lab.numel(), self.num_classes, self._cm,
                                _engine.stream_handle(self._dev))

  @property
  def total_cm(self):
    return self._cm.cpu().numpy()

  @property
  def ignored(self):
    """Pixels whose label or prediction was outside [0, num_classes) and therefore not counted."""
    return int(self._seen - int(self._cm.sum().item()))

  def result(self):
    cm = self.total_cm.astype(np.float64)
    tp = np.diag(cm)
    denom = cm.sum(axis=0) + cm.sum(axis=1) - tp
    valid = denom != 0
    if not valid.any():
      return 0.0
    return float(np.where(valid, tp / np.where(valid, denom, 1), 0.0).sum() / valid.sum())


def masked_labels(samples, mask, none_index):
  """The reference's ``label[~mask] = config.CLASSES.index("None")`` (eval.py / inference.py:65-66,
  data_loader/data_loader.py:176-180) on the device: samples [N,H,W,6] float32 (channel 5 = label),
  mask [N,H,W] uint8/bool -> int32 labels [N,H,W]."""
  import torch
  none = torch.tensor(none_index, dtype=torch.int32, device=samples.device)
  return torch.where(mask.bool(), samples[..., 5].to(torch.int32), none)


def evaluation(arg):
  if arg.path_to_model:
    model = load_model(arg.path_to_model, model_name=arg.model, config_name=arg.config)   # .npz or SavedModel dir
    config = model.mc
  else:
    config, model = load_model_config(arg.model, arg.config)
    model.init_weights(4321)
  files = sorted(glob.glob(os.path.join(arg.data_path, arg.image_set, "*.npy")))
  if not files:
    raise SystemExit("no scans under %s" % os.path.join(arg.data_path, arg.image_set))
  none_index = config.CLASSES.index("None")
  miou_tracker = MeanIoU(num_classes=config.NUM_CLASS, name="MeanIoU")
  print("Performing Evaluation")
  import torch
  dev = _engine.torch_device(model.device)
  for b0 in range(0, len(files), arg.batch):
    samples = torch.from_numpy(np.stack([np.load(f).astype(np.float32) for f in files[b0:b0 + arg.batch]])).to(dev)
    predictions, mask = model.predict_raw(samples[..., :5], return_mask=True)       # device tensors
    label = masked_labels(samples, mask, none_index)
    miou_tracker.update_state(label, predictions)
  if miou_tracker.ignored:
    print("warning: %d pixels carry a label outside [0, %d) and were not counted"
          % (miou_tracker.ignored, config.NUM_CLASS))
  iou, recall, precision = confusion_matrix_to_iou_recall_precision(miou_tracker.total_cm)
  for i, cls in enumerate(config.CLASSES):
    print(cls.upper())
    print("IoU:       " + str(iou[i]))
    print("Recall:    " + str(recall[i]))
    print("Precision: " + str(precision[i]))
    print("")
  print("MIoU: {} ".format(miou_tracker.result()))
  return iou, recall, precision, miou_tracker.result()


def main(argv=None):
  parser = argparse.ArgumentParser(description="Parse Flags for the evaluation script!")
  parser.add_argument("-d", "--data_path", type=str, required=True, help="Absolute path to the dataset")
  parser.add_argument("-i", "--image_set", type=str, default="val",
                      help="Default: `val`. But can also be train, val or test")
  parser.add_argument("-t", "--eval_dir", type=str, default=None, help="accepted for compatibility with the reference; unused (no TensorBoard logs are written)")
  parser.add_argument("-p", "--path_to_model", type=str, default=None, help="Path to the model: .npz file or reference SavedModel directory")
  parser.add_argument("-m", "--model", type=str, default="squeezesegv2",
                      help="Model name either `squeezesegv2`, `darknet53`, `darknet21`")
  parser.add_argument("-n", "--config", type=str, default="squeezesegv2",
                      help="Which configuration: `squeezesegv2`, `squeezesegv2kitti`, ...")
  parser.add_argument("--batch", type=int, default=32)
  return evaluation(parser.parse_args(argv))


if __name__ == "__main__":
  main()
