"""Functional-simulator mode of the GPU suite (test infrastructure, like oracle/).

    PCLSEG_SIM=1 python -m pytest tests -m gpu          # the GPU tests on sim/_build/libpclseg_sim.so
    PCLSEG_SIM=cand ...                                  # the candidate kernel variants (make candidates; -DPCLSEG_CAND)
    PCLSEG_SIM=asan ...                                  # AddressSanitizer build (needs LD_PRELOAD, see sim/Makefile)

sim/ compiles the UNMODIFIED sources of pclsegmentation_amd/csrc for x86-64 against a stand-in <hip/hip_runtime.h>
(fibers for HIP threads, wave operations evaluated per 64-lane wave, LDS per block): the wave-level algorithms and
the host code run, and are checked, without an MI355X.  What that is worth and what it is not is stated in
sim/hip/hip_runtime.h.  The product never loads a simulator library: engine.py takes another binary only under
PCLSEG_DEBUG=1 + PCLSEG_LIB, which this module sets for the test process, and "device" tensors are CPU tensors
(the simulator's device memory is host memory) through the three seams engine.torch_device / stream_handle /
on_device."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_DIR = os.path.join(ROOT, "sim")
TARGETS = {"1": ("all", "libpclseg_sim.so"), "cand": ("cand", "libpclseg_sim_cand.so"), "asan": ("asan", "libpclseg_sim_asan.so"), "ubsan": ("ubsan", "libpclseg_sim_ubsan.so"),
           "traffic": ("traffic", "libpclseg_sim_traffic.so")}


def library(variant="1"):
  """Path of the simulator build `variant`, (re)built by make when a source is newer."""
  if os.environ.get("PCLSEG_SIM_LIB"):      # a simulator library built elsewhere (A/B of two simulator builds)
    return os.environ["PCLSEG_SIM_LIB"]
  target, name = TARGETS[variant]
  subprocess.check_call(["make", "-s", "-C", SIM_DIR, target])
  return os.path.join(SIM_DIR, "_build", name)


def activate(variant="1"):
  """Point engine.py at the simulator library and make CPU tensors stand in for device tensors.
  Must run before pclsegmentation_amd.engine is imported."""
  os.environ["PCLSEG_LIB"] = library(variant)
  os.environ["PCLSEG_DEBUG"] = "1"
  import torch
  from pclsegmentation_amd import engine as E
  assert E.DEBUG_LIB and E.LIB_PATH == os.environ["PCLSEG_LIB"]
  E.torch_device = lambda index=0: torch.device("cpu")
  E.stream_handle = lambda device=None: 0
  # every torch tensor is "device" memory, except the ones host_tensor() below handed out; NumPy arrays are host
  E.on_device = lambda x: hasattr(x, "data_ptr") and _host_range(x.data_ptr()) is None
  E.is_pinned = lambda x: bool(_host_range(x.data_ptr()))
  # `tensor.cpu()` on a device tensor is a copy ordered after the work of torch's current stream (the null stream here), and
  # torch.cuda.synchronize() waits for the device: in the simulator's lazy stream mode (HIPSIM_STREAMS=lazy) that is what makes
  # the queued kernels run — the library itself must have made the caller's stream wait for its lane streams
  lib = E.load_library()
  if hasattr(lib, "hipsim_sync_stream"):
    lib.hipsim_sync_stream.argtypes = [ctypes.c_void_p]
    lib.hipsim_sync_stream.restype = None
    lib.hipsim_sync_device.restype = None
    orig_cpu = torch.Tensor.cpu

    def cpu(self, *a, **k):
      if E.on_device(self):
        lib.hipsim_sync_stream(None)
      return orig_cpu(self, *a, **k)
    torch.Tensor.cpu = cpu
    global sync_device
    sync_device = lib.hipsim_sync_device
  return E


def sync_device():
  """torch.cuda.synchronize() of the simulated device (re-bound by activate())."""


_HOST = []     # (first byte, one past the last, pinned?) of every tensor host_tensor() handed out; views of them count
_KEEP = []     # the tensors themselves: a freed one's address would be handed to a later "device" tensor, which _host_range
               # would then call host memory (seen as a load-dependent failure of whichever test allocated next)


def _host_range(ptr):
  """-> pinned? of the host allocation that contains ptr, None if it is "device" memory."""
  for lo, hi, pinned in _HOST:
    if lo <= ptr < hi:
      return pinned
  return None


def host_tensor(shape, dtype, pinned):
  """A CPU tensor that the simulator-mode seams treat as HOST memory; pinned ones come from pclseg_host_alloc (the
  simulator's registry of page-locked memory, which hipHostGetDevicePointer / hipPointerGetAttributes consult)."""
  import ctypes
  import numpy as np
  import torch
  from pclsegmentation_amd import engine as E
  t = torch.empty(shape, dtype=dtype)
  if pinned:
    nbytes = max(1, t.numel() * t.element_size())
    p = E.load_library().pclseg_host_alloc(nbytes)
    assert p, "pclseg_host_alloc failed"
    buf = (ctypes.c_uint8 * nbytes).from_address(p)     # (never freed: a test process)
    t = torch.from_numpy(np.frombuffer(buf, dtype=np.uint8)).view(dtype)[:t.numel()].view(shape)
  _HOST.append((t.data_ptr(), t.data_ptr() + max(1, t.numel() * t.element_size()), bool(pinned)))
  _KEEP.append(t)
  return t
