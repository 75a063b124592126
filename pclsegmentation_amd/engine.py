"""ctypes binding of libpclseg.so (C ABI declared in include/pclseg.h).

The library is the product: there is no Python/NumPy/PyTorch implementation of any
network operation in this package, and no CPU fallback.  If the shared library is
missing or no MI355X is visible, construction fails with a RuntimeError.

PyTorch is used by callers only as a device-memory container: tensors are handed over as
raw ``data_ptr()`` device pointers together with the current HIP stream.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpclseg.so")
# A/B of two builds (scripts/, tests/simlib.py): another binary is loaded ONLY when the debug switch is set as well — a
# stray PCLSEG_LIB in a production environment changes nothing — and the swap is announced on stderr at load.
DEBUG_LIB = bool(os.environ.get("PCLSEG_LIB")) and os.environ.get("PCLSEG_DEBUG") == "1"
if DEBUG_LIB:
  LIB_PATH = os.environ["PCLSEG_LIB"]

OK = 0
ERR_BAD_ARG, ERR_BAD_SHAPE, ERR_MISSING_WEIGHT, ERR_HIP, ERR_OOM, ERR_STATE, ERR_RANGE, ERR_INTERNAL = -1, -2, -3, -4, -5, -6, -7, -8
MEM_HOST, MEM_DEVICE, MEM_HOST_ASYNC = 0, 1, 2
FLAG_KEEP_ACTIVATIONS = 1
FLAG_EXACT_F32 = 2
FLAG_RANGE_FALLBACK = 4
MATH_F16X3, MATH_F32 = 0, 1
MATH = {"f16x3": MATH_F16X3, "f32": MATH_F32}

ARCH_IDS = {"squeezesegv2": 0, "darknet21": 1, "darknet53": 2}
ACT = {"none": 0, "relu": 1, "leaky": 2, "sigmoid": 3}

# every symbol include/pclseg.h declares (checked by tests/test_host.py::test_library_exports_every_declared_symbol)
EXPORTS = [
  "pclseg_version", "pclseg_build_sha", "pclseg_last_error", "pclseg_plan", "pclseg_plan_ops", "pclseg_create", "pclseg_destroy",
  "pclseg_num_weights", "pclseg_weight_info", "pclseg_set_weight", "pclseg_finalize",
  "pclseg_packed_size", "pclseg_export_packed", "pclseg_import_packed",
  "pclseg_set_stream", "pclseg_sync", "pclseg_host_alloc", "pclseg_host_free", "pclseg_forward", "pclseg_forward_raw",
  "pclseg_num_tensors", "pclseg_tensor_info", "pclseg_read_tensor", "pclseg_op_normalize",
  "pclseg_op_conv2d", "pclseg_op_conv2d_transpose", "pclseg_op_max_pool", "pclseg_op_head",
  "pclseg_op_confusion_matrix", "pclseg_op_project", "pclseg_op_project_ex", "pclseg_op_split_f16_roundtrip",
]


class Desc(ctypes.Structure):
  _fields_ = [("arch", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32),
              ("num_class", ctypes.c_int32), ("none_index", ctypes.c_int32),
              ("output_stride", ctypes.c_int32), ("device", ctypes.c_int32),
              ("micro_batch", ctypes.c_int32), ("flags", ctypes.c_uint32),
              ("mean", ctypes.c_double * 5), ("std", ctypes.c_double * 5)]


PROJ_ROW_FOV, PROJ_ROW_RING = 0, 1
PROJ_COL_FULL, PROJ_COL_FRONT = 0, 1
PROJ_NEAREST, PROJ_LAST = 0, 1


class ProjDesc(ctypes.Structure):
  _fields_ = [("h", ctypes.c_int32), ("w", ctypes.c_int32), ("row_mode", ctypes.c_int32),
              ("col_mode", ctypes.c_int32), ("winner", ctypes.c_int32), ("out_channels", ctypes.c_int32),
              ("fov_up", ctypes.c_float), ("fov_down", ctypes.c_float), ("empty", ctypes.c_float),
              ("left_phi", ctypes.c_double), ("right_phi", ctypes.c_double)]


class PlanInfo(ctypes.Structure):
  _fields_ = [("num_ops", ctypes.c_int32), ("num_weights", ctypes.c_int32),
              ("num_tensors", ctypes.c_int32), ("micro_batch", ctypes.c_int32),
              ("num_params", ctypes.c_int64), ("alg_macs_per_scan", ctypes.c_int64),
              ("alg_bytes_per_scan", ctypes.c_int64), ("workspace_bytes", ctypes.c_int64),
              ("packed_weight_bytes", ctypes.c_int64)]


_lib = None


def load_library():
  """dlopen libpclseg.so and declare signatures.  Raises RuntimeError if it is not built."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(
      "libpclseg.so not found at %s — build it with `make` (or __graft_entry__.build()); "
      "this engine has no CPU fallback" % LIB_PATH)
  lib = ctypes.CDLL(LIB_PATH)
  if DEBUG_LIB:   # A/B against an older build: entry points it lacks fail at the call, not at load
    import sys
    sys.stderr.write("pclseg: PCLSEG_DEBUG=1, loading %s instead of the shipped library\n" % LIB_PATH)
    class _Missing:
      def __init__(self, name):
        self.name, self.argtypes, self.restype = name, None, None
      def __call__(self, *a):
        raise RuntimeError("%s is not exported by %s" % (self.name, LIB_PATH))
    for name in EXPORTS:
      if not hasattr(lib, name):
        setattr(lib, name, _Missing(name))
  vp, i32, f32p = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
  lib.pclseg_version.restype = ctypes.c_int
  lib.pclseg_build_sha.argtypes = []
  lib.pclseg_last_error.restype = ctypes.c_char_p
  lib.pclseg_last_error.argtypes = [vp]
  lib.pclseg_plan.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(PlanInfo)]
  lib.pclseg_plan_ops.argtypes = [ctypes.POINTER(Desc), ctypes.c_char_p, ctypes.c_size_t]
  lib.pclseg_create.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(vp)]
  lib.pclseg_destroy.argtypes = [vp]
  lib.pclseg_num_weights.argtypes = [vp]
  lib.pclseg_weight_info.argtypes = [vp, i32, ctypes.c_char_p, ctypes.c_size_t,
                                     ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
  lib.pclseg_set_weight.argtypes = [vp, ctypes.c_char_p, vp, ctypes.POINTER(ctypes.c_int64), i32]
  lib.pclseg_finalize.argtypes = [vp]
  lib.pclseg_packed_size.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
  lib.pclseg_export_packed.argtypes = [vp, vp, ctypes.c_size_t, i32]
  lib.pclseg_import_packed.argtypes = [vp, vp, ctypes.c_size_t, i32]
  lib.pclseg_set_stream.argtypes = [vp, vp]
  lib.pclseg_sync.argtypes = [vp]
  lib.pclseg_host_alloc.argtypes = [ctypes.c_size_t]
  lib.pclseg_host_free.argtypes = [vp]
  lib.pclseg_forward.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32]
  lib.pclseg_forward_raw.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32]
  lib.pclseg_num_tensors.argtypes = [vp]
  lib.pclseg_tensor_info.argtypes = [vp, i32, ctypes.c_char_p, ctypes.c_size_t,
                                     ctypes.POINTER(ctypes.c_int64)]
  lib.pclseg_read_tensor.argtypes = [vp, i32, vp, ctypes.c_size_t]
  d5 = ctypes.POINTER(ctypes.c_double)
  lib.pclseg_op_normalize.argtypes = [vp, i32, i32, i32, d5, d5, vp, vp]
  lib.pclseg_op_conv2d.argtypes = [vp, i32, i32, i32, i32, f32p, i32, i32, i32, i32, f32p, f32p,
                                   f32p, f32p, f32p, i32, vp, vp, i32]
  lib.pclseg_op_conv2d_transpose.argtypes = [vp, i32, i32, i32, i32, f32p, i32, f32p, f32p, f32p,
                                             f32p, f32p, i32, vp, i32]
  lib.pclseg_op_max_pool.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp]
  lib.pclseg_op_head.argtypes = [vp, vp, i32, i32, i32, i32, f32p, f32p, i32, i32, vp, vp, vp, i32]
  lib.pclseg_op_confusion_matrix.argtypes = [vp, vp, ctypes.c_size_t, i32, vp, vp]
  lib.pclseg_op_project.argtypes = [vp, ctypes.c_size_t, i32, i32, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, vp, vp, vp, vp]
  lib.pclseg_op_project_ex.argtypes = [ctypes.POINTER(ProjDesc), vp, i32, ctypes.c_size_t, vp, vp, vp, vp, i32,
                                       vp, vp, vp, vp]
  lib.pclseg_op_split_f16_roundtrip.argtypes = [f32p, i32, i32, i32, i32, vp, vp]
  for name in EXPORTS:
    fn = getattr(lib, name)
    if name not in ("pclseg_last_error", "pclseg_host_alloc", "pclseg_build_sha"):
      fn.restype = ctypes.c_int
  lib.pclseg_host_alloc.restype = ctypes.c_void_p
  lib.pclseg_build_sha.restype = ctypes.c_char_p
  _lib = lib
  return lib


def build_sha():
  """Source hash baked into the loaded binary (Makefile -DPCLSEG_SRC_SHA), "unknown" for a foreign build."""
  return load_library().pclseg_build_sha().decode()


def _raise(rc, handle=None):
  """Map a negative status to the exception the reference's Keras path would raise:
  shape problems -> ValueError, everything else -> RuntimeError."""
  lib = load_library()
  msg = lib.pclseg_last_error(handle).decode("utf-8", "replace")
  if rc in (ERR_BAD_SHAPE, ERR_BAD_ARG):
    raise ValueError("pclseg: %s" % msg)
  if rc == ERR_MISSING_WEIGHT:
    raise KeyError("pclseg: %s" % msg)
  if rc == ERR_OOM:
    raise MemoryError("pclseg: %s" % msg)
  if rc == ERR_RANGE:
    raise FloatingPointError("pclseg: %s" % msg)
  raise RuntimeError("pclseg (status %d): %s" % (rc, msg))


def check(rc, handle=None):
  if rc != OK:
    _raise(rc, handle)
  return rc


def make_desc(arch, height, width, num_class, none_index, mean, std, output_stride=16, device=0,
              micro_batch=0, flags=0):
  d = Desc()
  d.arch = ARCH_IDS[arch] if isinstance(arch, str) else int(arch)
  d.height, d.width = int(height), int(width)
  d.num_class, d.none_index = int(num_class), int(none_index)
  d.output_stride, d.device = int(output_stride), int(device)
  d.micro_batch, d.flags = int(micro_batch), int(flags)
  mean = np.asarray(mean, np.float64).reshape(5)
  std = np.asarray(std, np.float64).reshape(5)
  for i in range(5):
    d.mean[i] = float(mean[i])
    d.std[i] = float(std[i])
  return d


def plan(desc):
  """CPU-only graph description (ops, params, algorithmic MACs/bytes, workspace)."""
  lib = load_library()
  info = PlanInfo()
  check(lib.pclseg_plan(ctypes.byref(desc), ctypes.byref(info)))
  return {k: getattr(info, k) for k, _ in PlanInfo._fields_}


def plan_ops(desc):
  """CPU-only: names of the kernel launches of one micro-batch, in launch order."""
  buf = ctypes.create_string_buffer(1 << 16)
  check(load_library().pclseg_plan_ops(ctypes.byref(desc), buf, len(buf)))
  return [line.split("\t")[0] for line in buf.value.decode().splitlines()]


def plan_op_macs(desc):
  """CPU-only: [(launch name, multiply-accumulates per scan)] in launch order."""
  buf = ctypes.create_string_buffer(1 << 16)
  check(load_library().pclseg_plan_ops(ctypes.byref(desc), buf, len(buf)))
  return [(l.split("\t")[0], int(l.split("\t")[1])) for l in buf.value.decode().splitlines()]


def plan_op_resources(desc):
  """CPU-only: [(launch name, MACs per scan, LDS bytes per block, threads per block, blocks per scan)]."""
  buf = ctypes.create_string_buffer(1 << 16)
  check(load_library().pclseg_plan_ops(ctypes.byref(desc), buf, len(buf)))
  return [(f[0],) + tuple(int(x) for x in f[1:5]) for f in (l.split("\t") for l in buf.value.decode().splitlines())]


# ---- where tensors live.  PyTorch is only the device-memory container of this package; these three functions are
# the one place that says how a device, its current stream, "is on the device" and "is page-locked" are spelled in torch.
def torch_device(index=0):
  """torch.device of HIP device ``index`` (tensors handed to MEM_DEVICE calls live there)."""
  import torch
  return torch.device("cuda", int(index))


def stream_handle(device):
  """Raw hipStream_t of torch's current stream on ``device`` (what pclseg_set_stream takes)."""
  import torch
  return torch.cuda.current_stream(device).cuda_stream


def on_device(x):
  """True for a tensor in device memory; NumPy arrays and CPU tensors are host memory."""
  return bool(getattr(x, "is_cuda", False))


def is_pinned(x):
  """True for a page-locked host tensor (MEM_HOST_ASYNC needs one)."""
  return bool(x.is_pinned())


def _host_f32(a):
  return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(x):
  """Raw address of a NumPy array, a torch tensor, an int, or None."""
  if x is None:
    return None
  if isinstance(x, int):
    return ctypes.c_void_p(x)
  if isinstance(x, np.ndarray):
    return ctypes.c_void_p(x.ctypes.data)
  return ctypes.c_void_p(x.data_ptr())  # torch.Tensor


_NP_OF_TORCH = {"torch.float32": np.float32, "torch.int32": np.int32, "torch.uint8": np.uint8,
                "torch.bool": np.uint8, "torch.int64": np.int64}


def _checked(x, dtype, count, what, mem=None):
  """Validate a buffer handed to the C ABI: dtype, C-contiguity, element count and, when ``mem`` is
  given, that it lives on the side the call says (host array / CPU tensor vs CUDA tensor).  The
  library reads ``count`` elements of ``dtype`` from the raw address: a float64 or strided array, or
  a short buffer, would otherwise be read or written as garbage."""
  if x is None:
    return None
  want = np.dtype(dtype)
  if isinstance(x, np.ndarray):
    if x.dtype != want or not x.flags["C_CONTIGUOUS"]:
      raise ValueError("pclseg: %s must be a C-contiguous %s array, got %s%s" % (
        what, want.name, x.dtype, "" if x.flags["C_CONTIGUOUS"] else " (not contiguous)"))
    if x.size < count:
      raise ValueError("pclseg: %s holds %d elements, the call needs %d" % (what, x.size, count))
    if mem == MEM_DEVICE:
      raise ValueError("pclseg: %s is a host array but the call was made with mem=MEM_DEVICE" % what)
    return x
  if hasattr(x, "data_ptr"):   # torch.Tensor
    got = _NP_OF_TORCH.get(str(x.dtype))
    if got is None or np.dtype(got).itemsize != want.itemsize or (np.dtype(got).kind == "f") != (want.kind == "f"):
      raise ValueError("pclseg: %s must be a %s tensor, got %s" % (what, want.name, x.dtype))
    if not x.is_contiguous():
      raise ValueError("pclseg: %s must be contiguous" % what)
    if x.numel() < count:
      raise ValueError("pclseg: %s holds %d elements, the call needs %d" % (what, x.numel(), count))
    if mem == MEM_HOST_ASYNC and not on_device(x) and not is_pinned(x):
      raise ValueError("pclseg: %s must be page-locked (pin_memory) for MEM_HOST_ASYNC" % what)
    if mem is not None and on_device(x) != (mem == MEM_DEVICE):
      raise ValueError("pclseg: %s lives on %s but the call was made with mem=%s" % (
        what, "the device" if on_device(x) else "the host", "MEM_DEVICE" if mem == MEM_DEVICE else "MEM_HOST[_ASYNC]"))
    return x
  return x   # raw integer address: the caller vouches for it


class Engine:
  """One model graph on one device (wraps a pclseg_handle)."""

  def __init__(self, desc):
    self.lib = load_library()
    self.desc = desc
    self._h = ctypes.c_void_p()
    check(self.lib.pclseg_create(ctypes.byref(desc), ctypes.byref(self._h)))
    self.finalized = False

  def close(self):
    if getattr(self, "_h", None):
      self.lib.pclseg_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  # -- weights
  def weight_inventory(self):
    out = []
    n = self.lib.pclseg_num_weights(self._h)
    name = ctypes.create_string_buffer(256)
    shape = (ctypes.c_int64 * 4)()
    ndim = ctypes.c_int()
    for i in range(n):
      check(self.lib.pclseg_weight_info(self._h, i, name, 256, shape, ctypes.byref(ndim)), self._h)
      out.append((name.value.decode(), tuple(int(shape[j]) for j in range(ndim.value))))
    return out

  def set_weight(self, path, array):
    a = _host_f32(array)
    shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
    check(self.lib.pclseg_set_weight(self._h, path.encode(), _ptr(a), shape, a.ndim), self._h)

  def set_weights(self, weights):
    for path, _ in self.weight_inventory():
      if path not in weights:
        raise KeyError("pclseg: weight set has no tensor '%s'" % path)
      self.set_weight(path, weights[path])

  def finalize(self):
    check(self.lib.pclseg_finalize(self._h), self._h)
    self.finalized = True

  # -- packed parameters (multi-GPU start-up: rank 0 exports, the others import the broadcast blob)
  def packed_size(self):
    n = ctypes.c_size_t()
    check(self.lib.pclseg_packed_size(self._h, ctypes.byref(n)), self._h)
    return int(n.value)

  @staticmethod
  def _blob(buf):
    """-> (address, bytes, mem) of a uint8 NumPy array or torch tensor (host or device)."""
    if isinstance(buf, np.ndarray):
      if buf.dtype != np.uint8 or not buf.flags["C_CONTIGUOUS"]:
        raise ValueError("pclseg: packed blob must be a C-contiguous uint8 array")
      return _ptr(buf), buf.size, MEM_HOST
    if str(buf.dtype) != "torch.uint8" or not buf.is_contiguous():
      raise ValueError("pclseg: packed blob must be a contiguous uint8 tensor")
    return _ptr(buf), buf.numel(), (MEM_DEVICE if on_device(buf) else MEM_HOST)

  def export_packed(self, buf):
    p, n, mem = self._blob(buf)
    check(self.lib.pclseg_export_packed(self._h, p, n, mem), self._h)

  def import_packed(self, buf):
    p, n, mem = self._blob(buf)
    check(self.lib.pclseg_import_packed(self._h, p, n, mem), self._h)
    self.finalized = True

  # -- execution
  def set_stream(self, stream_handle):
    check(self.lib.pclseg_set_stream(self._h, ctypes.c_void_p(stream_handle or 0)), self._h)

  def sync(self):
    check(self.lib.pclseg_sync(self._h), self._h)

  def _px(self, n):
    return int(n) * self.desc.height * self.desc.width

  def forward(self, lidar, mask, n, preds, probs=None, logits=None, mem=MEM_DEVICE):
    px, nc = self._px(n), self.desc.num_class
    _checked(lidar, np.float32, px * 6, "lidar", mem)
    _checked(mask, np.uint8, px, "mask", mem)
    _checked(preds, np.int32, px, "preds", mem)
    _checked(probs, np.float32, px * nc, "probs", mem)
    _checked(logits, np.float32, px * nc, "logits", mem)
    check(self.lib.pclseg_forward(self._h, _ptr(lidar), _ptr(mask), int(n), _ptr(preds),
                                  _ptr(probs), _ptr(logits), mem), self._h)

  def forward_raw(self, scans, n, preds, probs=None, logits=None, mask_out=None, mem=MEM_DEVICE):
    px, nc = self._px(n), self.desc.num_class
    _checked(scans, np.float32, px * 5, "scans", mem)
    _checked(preds, np.int32, px, "preds", mem)
    _checked(probs, np.float32, px * nc, "probs", mem)
    _checked(logits, np.float32, px * nc, "logits", mem)
    _checked(mask_out, np.uint8, px, "mask_out", mem)
    check(self.lib.pclseg_forward_raw(self._h, _ptr(scans), int(n), _ptr(preds), _ptr(probs),
                                      _ptr(logits), _ptr(mask_out), mem), self._h)

  # -- debug
  def tensors(self):
    out = []
    n = self.lib.pclseg_num_tensors(self._h)
    name = ctypes.create_string_buffer(256)
    shape = (ctypes.c_int64 * 4)()
    for i in range(n):
      check(self.lib.pclseg_tensor_info(self._h, i, name, 256, shape), self._h)
      out.append((name.value.decode(), tuple(int(s) for s in shape)))
    return out

  def read_tensor(self, index):
    name, shape = self.tensors()[index]
    buf = np.empty(shape, np.float32)
    check(self.lib.pclseg_read_tensor(self._h, index, _ptr(buf), buf.size), self._h)
    return buf


# ---- single-operator wrappers (device tensors in/out; used by the operator parity tests)
def op_normalize(scans_dev, n, h, w, mean, std, lidar6_dev, mask_dev):
  lib = load_library()
  m = (ctypes.c_double * 5)(*np.asarray(mean, np.float64).reshape(5))
  s = (ctypes.c_double * 5)(*np.asarray(std, np.float64).reshape(5))
  check(lib.pclseg_op_normalize(_ptr(scans_dev), n, h, w, m, s, _ptr(lidar6_dev), _ptr(mask_dev)))


def _opt(a):
  return None if a is None else _host_f32(a)


def op_conv2d(x_dev, n, h, w, cin, kernel, stride_w, bias, bn, act, residual_dev, y_dev, math="f16x3"):
  lib = load_library()
  k = _host_f32(kernel)
  kh, kw, kcin, cout = k.shape
  assert kcin == cin
  b = _opt(bias)
  g, be, mu, var = [_opt(v) for v in (bn if bn is not None else (None,) * 4)]
  check(lib.pclseg_op_conv2d(_ptr(x_dev), n, h, w, cin, _ptr(k), kh, kw, cout, stride_w, _ptr(b),
                             _ptr(g), _ptr(be), _ptr(mu), _ptr(var), ACT[act], _ptr(residual_dev),
                             _ptr(y_dev), MATH[math]))


def op_conv2d_transpose(x_dev, n, h, w, cin, kernel, bias, bn, act, y_dev, math="f16x3"):
  lib = load_library()
  k = _host_f32(kernel)
  assert k.shape[:2] == (1, 4) and k.shape[3] == cin
  cout = k.shape[2]
  b = _opt(bias)
  g, be, mu, var = [_opt(v) for v in (bn if bn is not None else (None,) * 4)]
  check(lib.pclseg_op_conv2d_transpose(_ptr(x_dev), n, h, w, cin, _ptr(k), cout, _ptr(b), _ptr(g),
                                       _ptr(be), _ptr(mu), _ptr(var), ACT[act], _ptr(y_dev), MATH[math]))


def op_max_pool(x_dev, n, h, w, c, k, stride_w, y_dev):
  check(load_library().pclseg_op_max_pool(_ptr(x_dev), n, h, w, c, k, stride_w, _ptr(y_dev)))


def op_head(x_dev, mask_dev, n, h, w, cin, kernel, bias, none_index, preds_dev, probs_dev=None,
            logits_dev=None, math="f16x3"):
  k = _host_f32(kernel)
  b = _host_f32(bias)
  nc = k.shape[3]
  check(load_library().pclseg_op_head(_ptr(x_dev), _ptr(mask_dev), n, h, w, cin, _ptr(k), _ptr(b),
                                      nc, none_index, _ptr(preds_dev), _ptr(probs_dev),
                                      _ptr(logits_dev), MATH[math]))


def op_split_f16_roundtrip(kernel):
  """CPU only: (recon float64, exponents int32) — the values the split-f16 fragments of a Keras conv
  kernel (kh,kw,Cin,Cout) represent after the per-output-channel pre-scale, and the exponents k."""
  k = _host_f32(kernel)
  kh, kw, cin, cout = k.shape
  recon = np.zeros(k.shape, np.float64)
  exps = np.zeros(cout, np.int32)
  check(load_library().pclseg_op_split_f16_roundtrip(_ptr(k), kh, kw, cin, cout, _ptr(recon), _ptr(exps)))
  return recon, exps


def op_confusion_matrix(labels_dev, preds_dev, count, num_class, cm_dev, stream=0):
  """cm[label][pred] += 1 (device int64 [NC,NC], accumulating)."""
  check(load_library().pclseg_op_confusion_matrix(_ptr(labels_dev), _ptr(preds_dev), int(count),
                                                  int(num_class), _ptr(cm_dev), ctypes.c_void_p(stream or 0)))


def op_project(points_dev, m, h, w, fov_up, fov_down, empty, image5_dev, proj_idx_dev, scratch_dev, stream=0):
  """Spherical projection of [m,4] points into an [h,w,5] range image (nearest point wins)."""
  check(load_library().pclseg_op_project(_ptr(points_dev), int(m), int(h), int(w), float(fov_up),
                                         float(fov_down), float(empty), _ptr(image5_dev),
                                         _ptr(proj_idx_dev), _ptr(scratch_dev), ctypes.c_void_p(stream or 0)))


def make_proj_desc(h, w, row_mode=PROJ_ROW_FOV, col_mode=PROJ_COL_FULL, winner=PROJ_NEAREST, out_channels=5,
                   fov_up=3.0, fov_down=-25.0, left_phi=0.0, right_phi=0.0, empty=0.0):
  d = ProjDesc()
  d.h, d.w, d.row_mode, d.col_mode, d.winner, d.out_channels = int(h), int(w), row_mode, col_mode, winner, out_channels
  d.fov_up, d.fov_down, d.empty = float(fov_up), float(fov_down), float(empty)
  d.left_phi, d.right_phi = float(left_phi), float(right_phi)
  return d


def op_project_ex(desc, points_dev, stride, m, ring_dev, depth_dev, labels_dev, lut_dev, image_dev, proj_idx_dev,
                  scratch_dev, stream=0):
  """Generalised projection (ring rows, front-view columns, last-point-wins, label / mask channels)."""
  check(load_library().pclseg_op_project_ex(ctypes.byref(desc), _ptr(points_dev), int(stride), int(m), _ptr(ring_dev),
                                            _ptr(depth_dev), _ptr(labels_dev), _ptr(lut_dev),
                                            0 if lut_dev is None else int(lut_dev.numel()), _ptr(image_dev),
                                            _ptr(proj_idx_dev), _ptr(scratch_dev), ctypes.c_void_p(stream or 0)))
