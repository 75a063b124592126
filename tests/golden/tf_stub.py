"""Shape-only stand-in for the TensorFlow names tests/golden/make_tf_golden.py uses — installed ONLY by its
`--check` dry mode, so the generator stays runnable (and tested) in an image without TensorFlow.  No
arithmetic: a Tensor is a shape; layers record their constructor arguments and how often they are called and
assert the input channels they were told to expect."""
import sys
import types


class Tensor:
  def __init__(self, shape):
    self.shape = tuple(int(s) for s in shape)

  def _same(self, other):
    if isinstance(other, (int, float)):     # scalar broadcast
      return Tensor(self.shape)
    assert isinstance(other, Tensor) and other.shape == self.shape, "shape mismatch %s vs %s" % (self.shape, getattr(other, "shape", other))
    return Tensor(self.shape)

  __add__ = __mul__ = _same


def _out_w(w, stride):
  return -(-w // stride)     # SAME padding


class Layer:
  def __init__(self, name=None, **kw):
    self.name, self.calls, self.expect_c = name, 0, None

  def __call__(self, x, *a, **kw):
    self.calls += 1
    if self.expect_c is not None:
      assert x.shape[-1] == self.expect_c, "%s: got %d input channels, the spec's kernel expects %d" % (type(self).__name__, x.shape[-1], self.expect_c)
    return self.compute(x)

  def compute(self, x):
    return Tensor(x.shape)


class Conv2D(Layer):
  def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", use_bias=True, **kw):
    super().__init__(**kw)
    self.filters, self.kernel_size, self.strides, self.padding, self.use_bias = filters, tuple(kernel_size), tuple(strides), padding, use_bias

  def compute(self, x):
    n, h, w, _ = x.shape
    return Tensor((n, _out_w(h, self.strides[0]), _out_w(w, self.strides[1]), self.filters))


class Conv2DTranspose(Conv2D):
  def compute(self, x):
    n, h, w, _ = x.shape
    return Tensor((n, h * self.strides[0], w * self.strides[1], self.filters))


class BatchNormalization(Layer):
  pass


class Softmax(Layer):
  def __init__(self, axis=-1, **kw):
    super().__init__(**kw)


class LeakyReLU(Layer):
  def __init__(self, alpha=0.3, **kw):
    super().__init__(**kw)
    self.alpha = alpha


class Model(Layer):
  def __call__(self, inputs, *a, **kw):
    return self.call(inputs, *a, **kw)


def _max_pool2d(x, ksize, strides, padding):
  assert padding == "SAME"
  s = strides if isinstance(strides, int) else strides[2]
  sh = 1 if not isinstance(strides, int) else strides
  n, h, w, c = x.shape
  return Tensor((n, _out_w(h, sh), _out_w(w, s), c))


def _concat(ts, axis):
  assert axis in (3, -1) and all(t.shape[:3] == ts[0].shape[:3] for t in ts)
  return Tensor(ts[0].shape[:3] + (sum(t.shape[3] for t in ts),))


def install():
  tf = types.ModuleType("tensorflow")
  tf.__version__ = "stub (shapes only)"
  layers = types.SimpleNamespace(Layer=Layer, Conv2D=Conv2D, Conv2DTranspose=Conv2DTranspose,
                                 BatchNormalization=BatchNormalization, Softmax=Softmax, LeakyReLU=LeakyReLU)
  tf.keras = types.SimpleNamespace(layers=layers, Model=Model)
  ident = lambda x: Tensor(x.shape)
  tf.nn = types.SimpleNamespace(relu=ident, sigmoid=ident, max_pool2d=_max_pool2d)
  tf.concat = _concat
  tf.constant = lambda a: a
  tf.int32 = "int32"
  tf.argmax = lambda x, axis=-1, output_type=None: Tensor(x.shape[:-1])
  tf.ones_like = ident
  tf.where = lambda c, a, b: a._same(c)._same(b)
  sys.modules["tensorflow"] = tf
  return tf
