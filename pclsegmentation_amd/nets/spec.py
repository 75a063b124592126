"""Weight inventories of the three networks, keyed by Keras attribute path.

Every tensor a trained reference model owns is listed here with the shape Keras gives
it, in layer-construction order:

* conv kernels ``(kh, kw, Cin, Cout)``, transposed-conv kernels ``(1, 4, Cout, Cin)``,
  ``bias (Cout,)``; each BatchNormalization = ``gamma, beta, moving_mean,
  moving_variance`` of shape ``(C,)``.
* SqueezeSegV2 — reference: nets/SqueezeSegV2.py:232-283 (CAM :46-64, FIRE :96-121,
  FIREUP :155-189).
* Darknet-21/53 — reference: nets/Darknet.py:187-260 (BasicBlock :34-52, EncoderLayer
  :76-94, DecoderLayer :110-128, block counts :142-145).

The same list drives the synthetic initialiser (nets/weights.py), the engine's weight
upload (engine.py) and the oracle, and it is cross-checked in tests against the list the
native graph builder reports through ``pclseg_weight_info``.
"""
from collections import namedtuple

WeightSpec = namedtuple("WeightSpec", "path shape kind fan_in")

DARKNET_BLOCKS = {21: (1, 1, 2, 2, 1), 53: (1, 2, 8, 8, 4)}


def _conv(path, kh, kw, cin, cout, bias=True):
  out = [WeightSpec(path + "/kernel", (kh, kw, cin, cout), "conv", kh * kw * cin)]
  if bias:
    out.append(WeightSpec(path + "/bias", (cout,), "bias", 0))
  return out


def _deconv(path, cin, cout):
  # Conv2DTranspose kernel layout is (kh, kw, Cout, Cin); each output pixel sums 2 taps.
  return [WeightSpec(path + "/kernel", (1, 4, cout, cin), "deconv", 2 * cin),
          WeightSpec(path + "/bias", (cout,), "bias", 0)]


def _bn(path, c):
  return [WeightSpec(path + "/gamma", (c,), "gamma", 0),
          WeightSpec(path + "/beta", (c,), "beta", 0),
          WeightSpec(path + "/moving_mean", (c,), "mean", 0),
          WeightSpec(path + "/moving_variance", (c,), "var", 0)]


def _cam(path, c, reduction=16):
  r = c // reduction
  return (_conv(path + "/squeeze", 1, 1, c, r) + _bn(path + "/squeeze_bn", r) +
          _conv(path + "/excitation", 1, 1, r, c) + _bn(path + "/excitation_bn", c))


def _fire(path, cin, sq, e1, e3, up=False):
  out = _conv(path + "/squeeze", 1, 1, cin, sq) + _bn(path + "/squeeze_bn", sq)
  if up:
    out += _deconv(path + "/upconv", sq, sq)
  out += _conv(path + "/expand1x1", 1, 1, sq, e1) + _bn(path + "/expand1x1_bn", e1)
  out += _conv(path + "/expand3x3", 3, 3, sq, e3) + _bn(path + "/expand3x3_bn", e3)
  return out


# (name, Cin, squeeze, expand1x1, expand3x3) — reference: nets/SqueezeSegV2.py:254-274
SSV2_FIRES = [("fire2", 64, 16, 64, 64), ("fire3", 128, 16, 64, 64),
              ("fire4", 128, 32, 128, 128), ("fire5", 256, 32, 128, 128),
              ("fire6", 256, 48, 192, 192), ("fire7", 384, 48, 192, 192),
              ("fire8", 384, 64, 256, 256), ("fire9", 512, 64, 256, 256)]
SSV2_FIREUPS = [("fire10", 512, 64, 128, 128), ("fire11", 256, 32, 64, 64),
                ("fire12", 128, 16, 32, 32), ("fire13", 64, 16, 32, 32)]


def squeezesegv2_spec(num_class, num_features=6):
  s = []
  s += _conv("conv1", 3, 3, num_features, 64) + _bn("bn1", 64)
  s += _cam("cam1", 64)
  s += _conv("conv1_skip", 1, 1, num_features, 64) + _bn("bn1_skip", 64)
  fires = dict((f[0], f) for f in SSV2_FIRES)
  for name in ("fire2",):
    s += _fire(*fires[name])
  s += _cam("cam2", 128)
  s += _fire(*fires["fire3"])
  s += _cam("cam3", 128)
  for name in ("fire4", "fire5", "fire6", "fire7", "fire8", "fire9"):
    s += _fire(*fires[name])
  for f in SSV2_FIREUPS:
    s += _fire(*f, up=True)
  s += _conv("conv14", 3, 3, 64, num_class)
  return s


def _basic_block(path, cin, planes):
  # 1x1 cin->planes[0], 3x3 planes[0]->planes[1], no biases (reference: nets/Darknet.py:34-52)
  return (_conv(path + "/conv1", 1, 1, cin, planes[0], bias=False) + _bn(path + "/bn1", planes[0]) +
          _conv(path + "/conv2", 3, 3, planes[0], planes[1], bias=False) + _bn(path + "/bn2", planes[1]))


def darknet_strides(output_stride):
  """Resolve the reference's "stride play" (nets/Darknet.py:158-181, :215-231)."""
  enc = [2, 2, 2, 2, 2]
  cur = 1
  for s in enc:
    cur *= s
  if output_stride <= cur:
    for i, stride in enumerate(reversed(enc)):
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


DARKNET_ENC_PLANES = [(32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)]
DARKNET_DEC_PLANES = [(1024, 512), (512, 256), (256, 128), (128, 64), (64, 32)]  # dec5..dec1


def darknet_spec(num_class, num_layers, output_stride=16, num_features=6):
  blocks = DARKNET_BLOCKS[num_layers]
  _, dec_strides = darknet_strides(output_stride)
  s = []
  s += _conv("conv1", 3, 3, num_features, 32, bias=False) + _bn("bn1", 32)
  for i, (planes, nb) in enumerate(zip(DARKNET_ENC_PLANES, blocks), start=1):
    p = "enc%d" % i
    s += _conv(p + "/conv1", 3, 3, planes[0], planes[1], bias=False) + _bn(p + "/bn1", planes[1])
    for j in range(nb):
      s += _basic_block("%s/residual_%d" % (p, j), planes[1], planes)
  for k, (planes, stride) in enumerate(zip(DARKNET_DEC_PLANES, dec_strides)):
    p = "dec%d" % (5 - k)
    if stride == 2:
      s += _deconv(p + "/upconv1", planes[0], planes[1])
    else:
      s += _conv(p + "/conv1", 3, 3, planes[0], planes[1], bias=True)
    s += _bn(p + "/bn1", planes[1])
    s += _basic_block(p + "/block", planes[1], planes)
  s += _conv("head", 3, 3, 32, num_class, bias=True)
  return s


def weight_spec(arch, num_class, num_layers=None, output_stride=16):
  arch = arch.lower()
  if arch == "squeezesegv2":
    return squeezesegv2_spec(num_class)
  if arch in ("darknet", "darknet21", "darknet53"):
    if num_layers is None:
      num_layers = int(arch[-2:])
    return darknet_spec(num_class, num_layers, output_stride)
  raise KeyError(arch)


def num_params(spec):
  n = 0
  for w in spec:
    k = 1
    for d in w.shape:
      k *= d
    n += k
  return n
