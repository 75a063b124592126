"""Second, independent CPU expression of the same graph with PyTorch (oneDNN) primitives.

THIS IS TEST INFRASTRUCTURE (see oracle/np_oracle.py's header; PARITY UNPINNED applies).
Two uses only:
  * tests cross-check it against the NumPy oracle (different primitives: F.conv2d with
    explicit asymmetric F.pad, F.conv_transpose2d, F.max_pool2d on -inf-padded input,
    F.batch_norm) so that a slip in either restatement shows up as a disagreement;
  * bench.py times it on the host cores as the "framework CPU" baseline standing in
    for the reference's TF2-CPU path, which cannot run here.
It must never be called from pclsegmentation_amd/.

Tensors are NCHW inside (PyTorch's native layout); inputs/outputs are NHWC like the
reference.  Reference lines are cited per function.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
_BLOCKS = {21: (1, 1, 2, 2, 1), 53: (1, 2, 8, 8, 4)}


def _same(size, k, s):
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  return total // 2, total - total // 2


class TorchNet:
  """Weights converted once to torch layouts; ``logits(lidar_nhwc)`` runs the graph."""

  def __init__(self, arch, weights, num_layers=None, output_stride=16, dtype=torch.float32):
    self.arch = arch.lower()
    self.dtype = dtype
    self.num_layers = num_layers if num_layers is not None else (
      int(self.arch[-2:]) if self.arch != "squeezesegv2" else None)
    self.output_stride = output_stride
    self.p = {}
    for k, v in weights.items():
      t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
      if k.endswith("/kernel"):
        if "upconv" in k:
          t = t.permute(3, 2, 0, 1).contiguous()   # (1,4,Co,Ci) -> (Ci,Co,1,4)
        else:
          t = t.permute(3, 2, 0, 1).contiguous()   # (kh,kw,Ci,Co) -> (Co,Ci,kh,kw)
      self.p[k] = t

  # ---- primitives
  def conv(self, x, path, stride_w=1):
    w = self.p[path + "/kernel"]
    b = self.p.get(path + "/bias")
    kh, kw = w.shape[2], w.shape[3]
    pt, pb = _same(x.shape[2], kh, 1)
    pl, pr = _same(x.shape[3], kw, stride_w)
    if pt or pb or pl or pr:
      x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, b, stride=(1, stride_w))

  def deconv(self, x, path):
    # TF SAME Conv2DTranspose (1,4)/(1,2): o = 2i + k - 1  ==  padding (0,1)
    return F.conv_transpose2d(x, self.p[path + "/kernel"], self.p[path + "/bias"],
                              stride=(1, 2), padding=(0, 1))

  def bn(self, x, path):
    return F.batch_norm(x, self.p[path + "/moving_mean"], self.p[path + "/moving_variance"],
                        self.p[path + "/gamma"], self.p[path + "/beta"], False, 0.0, BN_EPS)

  @staticmethod
  def pool(x, k, stride_w):
    pt, pb = _same(x.shape[2], k, 1)
    pl, pr = _same(x.shape[3], k, stride_w)
    x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(x, k, stride=(1, stride_w))

  # ---- SqueezeSegV2 (reference: nets/SqueezeSegV2.py:66-70, :123-127, :191-199, :285-323)
  def cam(self, x, p):
    s = F.relu(self.bn(self.conv(self.pool(x, 7, 1), p + "/squeeze"), p + "/squeeze_bn"))
    e = torch.sigmoid(self.bn(self.conv(s, p + "/excitation"), p + "/excitation_bn"))
    return x * e

  def fire(self, x, p, up=False):
    s = F.relu(self.bn(self.conv(x, p + "/squeeze"), p + "/squeeze_bn"))
    if up:
      s = F.relu(self.deconv(s, p + "/upconv"))
    e1 = F.relu(self.bn(self.conv(s, p + "/expand1x1"), p + "/expand1x1_bn"))
    e3 = F.relu(self.bn(self.conv(s, p + "/expand3x3"), p + "/expand3x3_bn"))
    return torch.cat([e1, e3], dim=1)

  def _ssv2(self, x_in):
    x = F.relu(self.bn(self.conv(x_in, "conv1", 2), "bn1"))
    cam1 = self.cam(x, "cam1")
    skip = self.bn(self.conv(x_in, "conv1_skip"), "bn1_skip")
    x = self.pool(cam1, 3, 2)
    x = self.fire(x, "fire2")
    x = self.cam(x, "cam2")
    x = self.fire(x, "fire3")
    cam3 = self.cam(x, "cam3")
    x = self.pool(cam3, 3, 2)
    x = self.fire(x, "fire4")
    fire5 = self.fire(x, "fire5")
    x = self.pool(fire5, 3, 2)
    for n in ("fire6", "fire7", "fire8", "fire9"):
      x = self.fire(x, n)
    x = self.fire(x, "fire10", True) + fire5
    x = self.fire(x, "fire11", True) + cam3
    x = self.fire(x, "fire12", True) + cam1
    x = self.fire(x, "fire13", True) + skip
    return self.conv(x, "conv14")

  # ---- Darknet (reference: nets/Darknet.py:54-66, :96-103, :130-138, :263-314)
  def block(self, x, p):
    y = F.leaky_relu(self.bn(self.conv(x, p + "/conv1"), p + "/bn1"), 0.1)
    y = F.leaky_relu(self.bn(self.conv(y, p + "/conv2"), p + "/bn2"), 0.1)
    return y + x

  def _darknet(self, x):
    os_ = self.output_stride
    # strides for OS in {8,16,32}: trailing encoder stages / leading decoder stages become 1
    n_ones = {32: 0, 16: 1, 8: 2, 4: 3, 2: 4, 1: 5}[os_]
    enc = [2] * (5 - n_ones) + [1] * n_ones
    dec = [1] * n_ones + [2] * (5 - n_ones)
    skips = []
    x = F.leaky_relu(self.bn(self.conv(x, "conv1"), "bn1"), 0.1)
    for i in range(5):
      p = "enc%d" % (i + 1)
      if enc[i] == 2:
        skips.append(x)
      x = F.leaky_relu(self.bn(self.conv(x, p + "/conv1", enc[i]), p + "/bn1"), 0.1)
      for j in range(_BLOCKS[self.num_layers][i]):
        x = self.block(x, "%s/residual_%d" % (p, j))
    for k in range(5):
      p = "dec%d" % (5 - k)
      if dec[k] == 2:
        x = self.deconv(x, p + "/upconv1")
      else:
        x = self.conv(x, p + "/conv1")
      x = F.leaky_relu(self.bn(x, p + "/bn1"), 0.1)
      x = self.block(x, p + "/block")
      if dec[k] == 2:
        x = x + skips.pop()
    return self.conv(x, "head")

  @torch.no_grad()
  def logits(self, lidar_nhwc):
    x = torch.as_tensor(np.asarray(lidar_nhwc)).to(self.dtype).permute(0, 3, 1, 2).contiguous()
    y = self._ssv2(x) if self.arch == "squeezesegv2" else self._darknet(x)
    return y.permute(0, 2, 3, 1).contiguous()

  @torch.no_grad()
  def __call__(self, lidar_nhwc, mask, none_index):
    """model([lidar, mask]) -> (probabilities, predictions) as numpy
    (reference: nets/SegmentationNetwork.py:58-69)."""
    lg = self.logits(lidar_nhwc)
    prob = torch.softmax(lg, dim=-1)
    pred = torch.argmax(prob, dim=-1).to(torch.int32)
    m = torch.as_tensor(np.asarray(mask, bool))
    pred = torch.where(m, pred, torch.full_like(pred, int(none_index)))
    return prob.numpy(), pred.numpy()
