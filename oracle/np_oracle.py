"""CPU oracle: NumPy restatement of the reference's forward pass.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it; the product path (pclsegmentation_amd/)
never does and fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference delegates all arithmetic to TensorFlow 2.9.1
(reference: requirements.txt:1), which is neither installed nor installable here, and
the reference ships no tests, golden outputs or trained weights.  This oracle is
therefore pinned only by (i) the hand-computed known-answer vectors of SURVEY.md
Appendix E (tests/test_oracle_kat.py), (ii) an independent PyTorch-CPU expression of
the same graph (oracle/torch_ref.py) and (iii) fixtures generated from itself.  TF was
never executed; a mis-read TF convention would go undetected.
PINNED BY THE REFERENCE ITSELF (its own NumPy code run in the build container, outputs committed as
fixtures): the pre-processing row — normalize_and_mask / parse_sample against DataLoader.parse_sample
(tests/golden/preproc_*.npz) — the spherical projections (tests/golden/projection*.npz) and the config
constants (tests/golden/configs.json).

Every function cites the reference lines it follows.  ``dtype`` selects float64 (the
reference arithmetic at higher precision, used as ground truth) or float32.
Layout is NHWC throughout, kernels are in Keras layout.
"""
import numpy as np

BN_EPS = 1e-3  # Keras BatchNormalization default; no epsilon= is passed anywhere in nets/*.py


# --------------------------------------------------------------------------- primitives
def same_pad(size, k, s):
  """TF SAME padding for one axis -> (out_size, pad_before, pad_after)."""
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  before = total // 2
  return out, before, total - before


def normalize_and_mask(raw, mean, std):
  """reference: inference.py:47-62 (twin: data_loader/data_loader.py:153-174).

  raw: [..., H, W, >=5] float (x, y, z, intensity, depth[, label]).  The reference
  first casts the file to float32 (:47), then evaluates (lidar - MEAN)/STD against
  float64 constants, so the arithmetic is float64 on float32-rounded inputs.
  Returns (lidar [...,H,W,6] float64, mask [...,H,W] bool).
  """
  lidar = np.asarray(raw)[..., :5].astype(np.float32)
  mask = lidar[..., 4] > 0
  lidar = (lidar - np.asarray(mean, np.float64).reshape(5)) / np.asarray(std, np.float64).reshape(5)
  lidar[~mask] = 0.0
  lidar = np.concatenate([lidar, mask[..., None].astype(lidar.dtype)], axis=-1)
  return lidar, mask


def parse_sample(sample, mean, std, none_index, cls_loss_weight=None):
  """reference: data_loader/data_loader.py:138-187 (DataLoader.parse_sample) and the same steps of
  inference.py:47-66 / eval.py's loop: the file content [..., H, W, 6] is cast to float32, normalised and
  masked (normalize_and_mask), and the label channel gets the "None" class wherever the mask is False
  (:176-180; inference.py:65-66).  Returns what parse_sample returns: lidar float32 [...,H,W,6], mask bool,
  label int32, weight float32 (class-wise loss weights, :182-185; zeros when no table is given).
  PINNED by tests/golden/preproc_*.npz, which are outputs of the reference function itself."""
  sample = np.asarray(sample).astype(np.float32)
  lidar, mask = normalize_and_mask(sample, mean, std)
  label = sample[..., 5].copy()
  label[~mask] = none_index
  weight = np.zeros(label.shape)
  if cls_loss_weight is not None:
    for l, wl in enumerate(np.asarray(cls_loss_weight).ravel()):
      weight[label == l] = wl
  return lidar.astype(np.float32), mask, label.astype(np.int32), weight.astype(np.float32)


def conv2d(x, kernel, bias=None, stride_w=1):
  """Keras Conv2D, padding SAME, strides (1, stride_w), cross-correlation.
  x [N,H,W,Cin], kernel (kh,kw,Cin,Cout).  Call sites: nets/SqueezeSegV2.py:46,56,96,
  105,114,155,173,182,232,243,276; nets/Darknet.py:34,44,76,121,187,255."""
  n, h, w, cin = x.shape
  kh, kw, kcin, cout = kernel.shape
  assert kcin == cin, (kcin, cin)
  ho, pt, pb = same_pad(h, kh, 1)
  wo, pl, pr = same_pad(w, kw, stride_w)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  out = np.zeros((n, ho, wo, cout), dtype=x.dtype)
  for i in range(kh):
    for j in range(kw):
      win = xp[:, i:i + ho, j:j + (wo - 1) * stride_w + 1:stride_w, :]
      out += np.matmul(win, kernel[i, j].astype(x.dtype))
  if bias is not None:
    out += bias.astype(x.dtype)
  return out


def conv2d_transpose_1x4_s2(x, kernel, bias=None):
  """Keras Conv2DTranspose(kernel_size=[1,4], strides=[1,2], padding SAME): W_out = 2 W_in,
  y[h,o,co] = b[co] + sum_{i,k: 2i+k-1=o} sum_ci x[h,i,ci] K[0,k,co,ci].
  Call sites: nets/SqueezeSegV2.py:165-171, nets/Darknet.py:113-118."""
  n, h, w, cin = x.shape
  kh, kw, cout, kcin = kernel.shape
  assert (kh, kw) == (1, 4) and kcin == cin
  full = np.zeros((n, h, 2 * w + 2, cout), dtype=x.dtype)   # o' = 2i + k, then crop 1 left
  for k in range(4):
    contrib = np.matmul(x, kernel[0, k].T.astype(x.dtype))  # [N,H,W,Cout]
    full[:, :, k:k + 2 * w:2, :] += contrib
  out = full[:, :, 1:1 + 2 * w, :]
  if bias is not None:
    out = out + bias.astype(x.dtype)
  return out


def batch_norm(x, gamma, beta, mean, var, eps=BN_EPS):
  """Keras BatchNormalization, training=False: (x-mean)*gamma/sqrt(var+eps)+beta."""
  t = x.dtype
  inv = gamma.astype(t) / np.sqrt(var.astype(t) + t.type(eps))
  return (x - mean.astype(t)) * inv + beta.astype(t)


def max_pool(x, k, stride_w=1):
  """MaxPool kxk, strides (1, stride_w), padding SAME; padded cells never win.
  Call sites: nets/SqueezeSegV2.py:40-44 (7x7 s1), :295,301,305 (3x3 s(1,2))."""
  n, h, w, c = x.shape
  ho, pt, pb = same_pad(h, k, 1)
  wo, pl, pr = same_pad(w, k, stride_w)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
  out = np.full((n, ho, wo, c), -np.inf, dtype=x.dtype)
  for i in range(k):
    for j in range(k):
      np.maximum(out, xp[:, i:i + ho, j:j + (wo - 1) * stride_w + 1:stride_w, :], out=out)
  return out


def relu(x):
  return np.maximum(x, 0)


def leaky_relu(x, alpha=0.1):
  """LeakyReLU(0.1) (reference: nets/Darknet.py:42,52,85,127,196)."""
  return np.where(x > 0, x, x * x.dtype.type(alpha))


def sigmoid(x):
  return 1.0 / (1.0 + np.exp(-x))


def softmax(x):
  """tf.keras.layers.Softmax(axis=-1): exp(x - max) / sum."""
  e = np.exp(x - x.max(axis=-1, keepdims=True))
  return e / e.sum(axis=-1, keepdims=True)


def segmentation_head(logits, mask, none_index):
  """reference: nets/SegmentationNetwork.py:58-69 — softmax, argmax over the
  PROBABILITIES (lowest index on ties, int32), invalid pixels -> index of "None"."""
  prob = softmax(logits)
  pred = np.argmax(prob, axis=-1).astype(np.int32)
  pred = np.where(mask, pred, np.int32(none_index)).astype(np.int32)
  return prob, pred


# --------------------------------------------------------------------------- weight access
class _W:
  """Weight lookup by Keras path prefix, cast to the working dtype."""

  def __init__(self, weights, dtype):
    self.w = weights
    self.t = np.dtype(dtype)

  def get(self, path):
    return np.asarray(self.w[path]).astype(self.t)

  def has(self, path):
    return path in self.w

  def conv(self, x, path, stride_w=1):
    b = self.get(path + "/bias") if self.has(path + "/bias") else None
    return conv2d(x, self.get(path + "/kernel"), b, stride_w)

  def deconv(self, x, path):
    return conv2d_transpose_1x4_s2(x, self.get(path + "/kernel"), self.get(path + "/bias"))

  def bn(self, x, path):
    return batch_norm(x, self.get(path + "/gamma"), self.get(path + "/beta"),
                      self.get(path + "/moving_mean"), self.get(path + "/moving_variance"))


# --------------------------------------------------------------------------- SqueezeSegV2
def cam(W, x, p):
  """CAM.call — reference: nets/SqueezeSegV2.py:66-70 (gate multiplies the un-pooled input)."""
  pool = max_pool(x, 7, 1)
  sq = relu(W.bn(W.conv(pool, p + "/squeeze"), p + "/squeeze_bn"))
  ex = sigmoid(W.bn(W.conv(sq, p + "/excitation"), p + "/excitation_bn"))
  return x * ex


def fire(W, x, p, up=False):
  """FIRE.call :123-127 / FIREUP.call :191-199 (deconv has ReLU but no BN)."""
  sq = relu(W.bn(W.conv(x, p + "/squeeze"), p + "/squeeze_bn"))
  if up:
    sq = relu(W.deconv(sq, p + "/upconv"))
  e1 = relu(W.bn(W.conv(sq, p + "/expand1x1"), p + "/expand1x1_bn"))
  e3 = relu(W.bn(W.conv(sq, p + "/expand3x3"), p + "/expand3x3_bn"))
  return np.concatenate([e1, e3], axis=3)


def squeezesegv2_logits(weights, lidar, dtype=np.float64, taps=None):
  """SqueezeSegV2.call up to the logits — reference: nets/SqueezeSegV2.py:285-323.
  ``taps``: optional dict that receives named intermediates."""
  W = _W(weights, dtype)
  x_in = np.asarray(lidar).astype(dtype)
  t = {} if taps is None else taps

  x = relu(W.bn(W.conv(x_in, "conv1", stride_w=2), "bn1"))                   # :289
  t["conv1"] = x
  cam1 = cam(W, x, "cam1")                                                    # :291
  t["cam1"] = cam1
  skip = W.bn(W.conv(x_in, "conv1_skip"), "bn1_skip")                        # :293
  t["conv1_skip"] = skip
  x = max_pool(cam1, 3, 2)                                                    # :295
  t["pool1"] = x
  x = fire(W, x, "fire2"); t["fire2"] = x
  x = cam(W, x, "cam2"); t["cam2"] = x
  x = fire(W, x, "fire3"); t["fire3"] = x
  cam3 = cam(W, x, "cam3"); t["cam3"] = cam3                                  # :299
  x = max_pool(cam3, 3, 2); t["pool3"] = x                                    # :301
  x = fire(W, x, "fire4"); t["fire4"] = x
  fire5 = fire(W, x, "fire5"); t["fire5"] = fire5                             # :303
  x = max_pool(fire5, 3, 2); t["pool5"] = x                                   # :305
  x = fire(W, x, "fire6"); t["fire6"] = x
  x = fire(W, x, "fire7"); t["fire7"] = x
  x = fire(W, x, "fire8"); t["fire8"] = x
  x = fire(W, x, "fire9"); t["fire9"] = x                                     # :309
  x = fire(W, x, "fire10", up=True) + fire5; t["fire10"] = x                 # :312-313
  x = fire(W, x, "fire11", up=True) + cam3; t["fire11"] = x                  # :314-315
  x = fire(W, x, "fire12", up=True) + cam1; t["fire12"] = x                  # :316-317
  x = fire(W, x, "fire13", up=True) + skip; t["fire13"] = x                  # :318-319
  logits = W.conv(x, "conv14")                                                # :323 (dropout = identity)
  t["logits"] = logits
  return logits


# --------------------------------------------------------------------------- Darknet
DARKNET_BLOCKS = {21: (1, 1, 2, 2, 1), 53: (1, 2, 8, 8, 4)}   # reference: nets/Darknet.py:142-145


def darknet_strides(output_stride):
  """reference: nets/Darknet.py:158-181 (encoder) and :215-231 (decoder)."""
  enc = [2, 2, 2, 2, 2]
  cur = 1
  for s in enc:
    cur *= s
  if not output_stride > cur:
    for i, stride in enumerate(reversed(enc), 0):
      if int(cur) != output_stride:
        if stride == 2:
          cur /= 2
          enc[-1 - i] = 1
        if int(cur) == output_stride:
          break
  dec = [2, 2, 2, 2, 2]
  cur = 1
  for s in dec:
    cur *= s
  for i, stride in enumerate(dec):
    if int(cur) != output_stride:
      if stride == 2:
        cur /= 2
        dec[i] = 1
      if int(cur) == output_stride:
        break
  return enc, dec


def basic_block(W, x, p):
  """BasicBlock.call — reference: nets/Darknet.py:54-66."""
  y = leaky_relu(W.bn(W.conv(x, p + "/conv1"), p + "/bn1"))
  y = leaky_relu(W.bn(W.conv(y, p + "/conv2"), p + "/bn2"))
  return y + x


def darknet_logits(weights, lidar, num_layers, output_stride=16, dtype=np.float64, taps=None):
  """Darknet.call up to the logits — reference: nets/Darknet.py:279-312, with the
  shape-driven skip bookkeeping of run_enc_block :263-269 / run_dec_block :271-277."""
  W = _W(weights, dtype)
  x = np.asarray(lidar).astype(dtype)
  t = {} if taps is None else taps
  enc_strides, dec_strides = darknet_strides(output_stride)
  blocks = DARKNET_BLOCKS[num_layers]
  skips, os_ = {}, 1

  def enc_step(x, y, os_):
    if y.shape[1] < x.shape[1] or y.shape[2] < x.shape[2]:
      skips[os_] = x
      os_ *= 2
    return y, os_

  y = W.conv(x, "conv1")                                                     # :288
  x, os_ = enc_step(x, y, os_)
  x = leaky_relu(W.bn(x, "bn1"))                                             # :289-290
  t["conv1"] = x
  for i in range(5):                                                         # :293-302
    p = "enc%d" % (i + 1)
    y = leaky_relu(W.bn(W.conv(x, p + "/conv1", stride_w=enc_strides[i]), p + "/bn1"))
    t[p + "/conv1"] = y
    for j in range(blocks[i]):
      y = basic_block(W, y, "%s/residual_%d" % (p, j))
      t["%s/residual_%d/conv2" % (p, j)] = y
    x, os_ = enc_step(x, y, os_)
    t[p] = x
  for k in range(5):                                                         # :305-309
    p = "dec%d" % (5 - k)
    if dec_strides[k] == 2:
      y = W.deconv(x, p + "/upconv1")
    else:
      y = W.conv(x, p + "/conv1")
    y = leaky_relu(W.bn(y, p + "/bn1"))
    t[p + ("/upconv1" if dec_strides[k] == 2 else "/conv1")] = y
    y = basic_block(W, y, p + "/block")
    if y.shape[2] > x.shape[2]:
      os_ //= 2
      y = y + skips[os_]
    x = y
    t[p] = x
    t[p + "/block/conv2"] = x
  logits = W.conv(x, "head")                                                 # :312
  t["logits"] = logits
  return logits


# --------------------------------------------------------------------------- model call
def forward(arch, weights, lidar, mask, none_index, num_layers=None, output_stride=16,
            dtype=np.float64, taps=None):
  """model([lidar, mask]) -> (probabilities, predictions, logits)
  (reference: nets/SegmentationNetwork.py:55-69; inference.py:75)."""
  arch = arch.lower()
  if arch == "squeezesegv2":
    logits = squeezesegv2_logits(weights, lidar, dtype, taps)
  else:
    if num_layers is None:
      num_layers = int(arch[-2:])
    logits = darknet_logits(weights, lidar, num_layers, output_stride, dtype, taps)
  prob, pred = segmentation_head(logits, np.asarray(mask, bool), none_index)
  return prob, pred, logits


# --------------------------------------------------------------------------- evaluation metrics
def confusion_matrix(labels, preds, num_class):
  """tf.metrics.MeanIoU.update_state (reference: eval.py:41-48): total_cm[label][pred] += 1.
  Exact integer counts; entries outside [0, num_class) are ignored."""
  l = np.asarray(labels).ravel().astype(np.int64)
  q = np.asarray(preds).ravel().astype(np.int64)
  ok = (l >= 0) & (l < num_class) & (q >= 0) & (q < num_class)
  cm = np.zeros((num_class, num_class), np.int64)
  np.add.at(cm, (l[ok], q[ok]), 1)
  return cm


def _divide_no_nan(a, b):
  return np.where(b != 0, a / np.where(b != 0, b, 1), 0.0)


def iou_recall_precision(cm):
  """reference: utils/util.py:64-79 (total_cm rows = labels, columns = predictions)."""
  cm = np.asarray(cm, np.float64)
  sum_over_col = cm.sum(axis=1)
  sum_over_row = cm.sum(axis=0)
  tp = np.diag(cm)
  fp = sum_over_row - tp
  fn = sum_over_col - tp
  return (_divide_no_nan(tp, tp + fp + fn), _divide_no_nan(tp, tp + fn), _divide_no_nan(tp, tp + fp))


def mean_iou(cm):
  """tf.metrics.MeanIoU.result: mean of the per-class IoU over classes that occur
  (denominator tp + fp + fn != 0); 0 if none does."""
  cm = np.asarray(cm, np.float64)
  tp = np.diag(cm)
  denom = cm.sum(axis=0) + cm.sum(axis=1) - tp
  valid = denom != 0
  return float((_divide_no_nan(tp, denom)).sum() / valid.sum()) if valid.any() else 0.0


# --------------------------------------------------------------------------- spherical projection
def range_projection(points, H, W, fov_up, fov_down):
  """Point cloud [M,4] (x,y,z,remission, float32) -> range image, restating
  dataset_convert/laserscan_semantic_kitti.py:106-166 (LaserScan.do_range_projection) with the
  same float32 NumPy operations in the same order.  Pinned by tests/golden/projection_*.npz,
  which are outputs of the reference's own code (tests/golden/make_projection_golden.py).
  Returns (proj_range [H,W], proj_xyz [H,W,3], proj_remission [H,W], proj_idx [H,W] int32);
  -1 marks pixels without a point; the nearest point wins a pixel."""
  points = np.asarray(points, np.float32)
  xyz, rem = points[:, :3], points[:, 3]
  fov_up_r = fov_up / 180.0 * np.pi
  fov_down_r = fov_down / 180.0 * np.pi
  fov = abs(fov_down_r) + abs(fov_up_r)
  depth = np.linalg.norm(xyz, 2, axis=1)
  yaw = -np.arctan2(xyz[:, 1], xyz[:, 0])
  pitch = np.arcsin(xyz[:, 2] / depth)
  proj_x = 0.5 * (yaw / np.pi + 1.0)
  proj_y = 1.0 - (pitch + abs(fov_down_r)) / fov
  proj_x *= W
  proj_y *= H
  proj_x = np.maximum(0, np.minimum(W - 1, np.floor(proj_x))).astype(np.int32)
  proj_y = np.maximum(0, np.minimum(H - 1, np.floor(proj_y))).astype(np.int32)
  order = np.argsort(depth)[::-1]          # decreasing depth: the nearest point is written last
  proj_range = np.full((H, W), -1, np.float32)
  proj_xyz = np.full((H, W, 3), -1, np.float32)
  proj_rem = np.full((H, W), -1, np.float32)
  proj_idx = np.full((H, W), -1, np.int32)
  py, px = proj_y[order], proj_x[order]
  proj_range[py, px] = depth[order]
  proj_xyz[py, px] = xyz[order]
  proj_rem[py, px] = rem[order]
  proj_idx[py, px] = np.arange(depth.shape[0])[order]
  return proj_range, proj_xyz, proj_rem, proj_idx


def range_projection_ring(points, ring_index, H, W):
  """Ring-index projection, restating dataset_convert/laserscan_nuscenes.py:191-223
  (LaserScan.do_range_projection_ring): column from the azimuth exactly as above, row =
  H-1-ring_index, NO depth ordering — plain fancy-index assignment in input order, so the LAST
  point of a pixel wins.  Pinned by tests/golden/projection2_ring_*.npz (outputs of the
  reference's own code).  Returns (proj_range, proj_xyz, proj_remission, proj_idx, proj_mask)."""
  points = np.asarray(points, np.float32)
  xyz, rem = points[:, :3], points[:, 3]
  depth = np.linalg.norm(xyz, 2, axis=1)
  yaw = -np.arctan2(xyz[:, 1], xyz[:, 0])
  proj_x = 0.5 * (yaw / np.pi + 1.0)
  proj_x *= W
  proj_x = np.maximum(0, np.minimum(W - 1, np.floor(proj_x))).astype(np.int32)
  proj_y = (H - 1) - np.asarray(ring_index, np.int32)
  proj_range = np.full((H, W), -1, np.float32)
  proj_xyz = np.full((H, W, 3), -1, np.float32)
  proj_rem = np.full((H, W), -1, np.float32)
  proj_idx = np.full((H, W), -1, np.int32)
  proj_range[proj_y, proj_x] = depth
  proj_xyz[proj_y, proj_x] = xyz
  proj_rem[proj_y, proj_x] = rem
  proj_idx[proj_y, proj_x] = np.arange(depth.shape[0])
  return proj_range, proj_xyz, proj_rem, proj_idx, (proj_idx > 0).astype(np.float32)


def label_projection(proj_idx, labels, learning_map=None):
  """SemLaserScan.do_label_projection (laserscan_nuscenes.py:377-383): labels gathered through
  proj_idx, pixels without a point keep 0; then, optionally, the converters' learning_map
  (dataset_convert/semantic_kitti.py:145,165: np.vectorize(dict.get) over the label image)."""
  out = np.zeros(proj_idx.shape, np.int32)
  mask = proj_idx >= 0
  out[mask] = np.asarray(labels)[proj_idx[mask]]
  if learning_map is not None:
    out = np.vectorize(learning_map.get)(out)
  return out


def information_map(pcl, H=32, W=240, left_phi=np.radians(24.32), right_phi=np.radians(22.23)):
  """Front-view map of preprocessing/convert_validation_pcd_to_npy.py:97-156: pcl [M,7] = x,y,z,i,
  ring,depth,label -> [H,W,7] float64 = x,y,z,i,d,label,mask; column = int((left_phi - atan2(y,x)) /
  ((right_phi+left_phi)/W)) (truncation toward zero), out-of-window points removed, row = H-1-ring,
  last point wins."""
  pcl = np.asarray(pcl, np.float64)
  x, y, z, i = pcl[:, 0], pcl[:, 1], pcl[:, 2], pcl[:, 3]
  r, d, l = pcl[:, 4].astype(int), pcl[:, 5], pcl[:, 6].astype(int)
  dphi = (right_phi + left_phi) / W
  col = ((left_phi - np.arctan2(y, x)) / dphi).astype(int)
  mask = np.zeros_like(l)
  mask[d > 0] = 1
  keep = ~np.logical_or(col < 0, col >= W)
  out = np.zeros((H, W, 7))
  rows = (H - 1) - r[keep]
  for c, v in enumerate((x, y, z, i, d, l, mask)):
    out[rows, col[keep], c] = v[keep]
  return out
