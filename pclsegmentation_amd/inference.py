"""Inference CLI with the reference's flags and outputs (reference: inference.py:36-132).

  python -m pclsegmentation_amd.inference -d './data/*.npy' -m squeezesegv2 -t out/ -p model.npz

Same four flags (``-d/--input_path`` glob, ``-m/--model``, ``-t/--output_dir``,
``-p/--path_to_model``) and the same three files per scan: ``pred_<name>.npy`` (int32 [H,W]),
``plot_<name>.png`` and ``plot_gt_<name>.png`` (RGBA, colours = 255 * CLS_COLOR_MAP[id]; the
reference blends with alpha = 1.0, so the images are the label colours).

Differences, all deliberate:
  * ``--model`` is honoured (the reference parses it and then hard-codes SqueezeSegV2Config) and
    ``--config`` selects the config (default: the model file's own, else the model's namesake);
  * scans are batched (``--batch``) and normalise+mask runs on the device inside
    ``pclseg_forward_raw`` instead of per-scan NumPy on the host;
  * the per-scan outputs (one .npy and two PNG encodes, which dominate the reference's wall
    time) are written by a small thread pool while the GPU runs the next batch;
  * ``<name>`` is the file's base name without extension (the reference's
    ``f.strip('.npy')`` strips characters, not the suffix);
  * ``-p`` takes this engine's ``.npz`` model file (``model.save``); without ``-p`` the model
    gets the seeded synthetic weights and says so.
"""
import argparse
import collections
import concurrent.futures
import glob
import os
import sys

import numpy as np

from . import load_model, load_model_config
from .utils.util import normalize


def _save_plots(out_dir, name, config, predictions, label, depth_feature):
  from PIL import Image
  cmap = np.asarray(config.CLS_COLOR_MAP, np.float64)
  depth_map = Image.fromarray((255 * normalize(depth_feature)).astype(np.uint8))
  for prefix, ids in (("plot_", predictions), ("plot_gt_", label)):
    label_map = Image.fromarray((255 * cmap[ids]).astype(np.uint8))
    blend = Image.blend(depth_map.convert("RGBA"), label_map.convert("RGBA"), alpha=1.0)
    blend.save(os.path.join(out_dir, prefix + name + ".png"))


def inference(arg):
  files = sorted(glob.glob(arg.input_path))
  if not files:
    raise SystemExit("no input files match %r" % arg.input_path)
  if arg.path_to_model:
    model = load_model(arg.path_to_model, model_name=arg.model, config_name=arg.config)   # .npz or SavedModel dir
    config = model.mc
    if arg.model and model.arch_name() != arg.model.lower() and not (
        arg.model.lower().startswith("darknet") and model.arch_name().startswith("darknet")):
      raise SystemExit("model file holds %s, --model says %s" % (model.arch_name(), arg.model))
  else:
    config, model = load_model_config(arg.model, arg.config or arg.model)
    model.init_weights(4321)
    print("no --path_to_model given: using seeded synthetic weights", file=sys.stderr)
  os.makedirs(arg.output_dir, exist_ok=True)
  none_index = config.CLASSES.index("None")

  try:
    import tqdm
    batches = tqdm.tqdm(range(0, len(files), arg.batch))
  except ImportError:
    batches = range(0, len(files), arg.batch)
  pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, arg.writers))
  pending = []

  def write_one(f, sample, pred, m):
    name = os.path.splitext(os.path.basename(f))[0]
    np.save(os.path.join(arg.output_dir, "pred_" + name + ".npy"), pred)
    if arg.no_plots:
      return
    if sample.shape[2] > 5:
      label = sample[:, :, 5].astype(np.int32)
      label[~m] = none_index                      # reference: inference.py:65-68
    else:
      label = np.full(pred.shape, none_index, np.int32)
    # the reference plots channel 3 of the NORMALISED lidar tensor (inference.py:88)
    feat = (sample[:, :, 3].astype(np.float64) - config.INPUT_MEAN.ravel()[3]) / config.INPUT_STD.ravel()[3]
    feat[~m] = 0.0
    _save_plots(arg.output_dir, name, config, pred, label, feat)

  # The reference np.loads its 6.3 MB float64 files one by one on the main thread (inference.py:44-47),
  # which dominates its wall time.  Here a pool of loader threads reads and casts the
  # NEXT batches while the GPU works on the current one (np.load and astype release the GIL); at most
  # `--prefetch` batches are held in memory.
  loaders = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, arg.loaders))

  def load_one(f):
    return np.load(f).astype(np.float32, copy=False)

  starts = list(range(0, len(files), arg.batch))
  ahead = collections.deque()

  def submit(i):
    if i < len(starts):
      ahead.append([loaders.submit(load_one, f) for f in files[starts[i]:starts[i] + arg.batch]])

  for i in range(max(1, arg.prefetch)):
    submit(i)
  for bi, b0 in enumerate(batches):
    chunk = files[b0:b0 + arg.batch]
    samples = [fut.result() for fut in ahead.popleft()]
    submit(bi + max(1, arg.prefetch))
    shapes = {s.shape[:2] for s in samples}
    if len(shapes) != 1:
      raise SystemExit("scans in one batch must share a shape, got %s" % sorted(shapes))
    raw = np.stack([s[:, :, :5] for s in samples])
    predictions, mask = model.predict_raw(raw, return_mask=True)
    predictions = predictions.numpy()
    for f, sample, pred, m in zip(chunk, samples, predictions, mask):
      print("Process: {0}".format(f))
      pending.append(pool.submit(write_one, f, sample, np.array(pred), np.array(m)))
  for fut in pending:
    fut.result()          # surface any writer exception
  pool.shutdown()
  loaders.shutdown()


def main(argv=None):
  parser = argparse.ArgumentParser(description="Parse Flags for the inference script!")
  parser.add_argument("-d", "--input_path", type=str, required=True,
                      help="Input LiDAR scans to be detected. Must be a glob pattern input such as "
                           "`./data/samples/*.npy` !")
  parser.add_argument("-m", "--model", type=str, default="squeezesegv2",
                      help="Model name either `squeezesegv2`, `darknet53`, `darknet21`")
  parser.add_argument("-t", "--output_dir", type=str, required=True,
                      help="Directory where to write the model predictions and visualizations")
  parser.add_argument("-p", "--path_to_model", type=str, default=None, help="Path to the model: .npz file or reference SavedModel directory")
  parser.add_argument("-c", "--config", type=str, default=None,
                      help="Config name (config_map key); default: the model's namesake")
  parser.add_argument("--batch", type=int, default=32, help="scans per forward call")
  parser.add_argument("--no_plots", action="store_true", help="write only pred_*.npy")
  parser.add_argument("--writers", type=int, default=4, help="output writer threads")
  parser.add_argument("--loaders", type=int, default=4, help="input loader threads (read + cast ahead of the GPU)")
  parser.add_argument("--prefetch", type=int, default=2, help="batches loaded ahead")
  inference(parser.parse_args(argv))


if __name__ == "__main__":
  main()
