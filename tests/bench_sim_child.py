"""Child process of tests/test_sim_only.py::test_bench_line_and_its_legs_on_the_simulator: bench.py's OWN code — main(),
the secondary rows, c1_gpu_leg, parity_check, cpu_baseline — executed end to end on the functional simulator with
workloads shrunk to sizes it finishes in seconds.  A smoke test of bench.py's control flow and of the JSON line's
shape (the timings it prints here mean nothing); it exists because bench.py must work the first time a driver runs
it on an MI355X.  torch.cuda is replaced by a thin stand-in for THIS process only."""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PCLSEG_SIM", "1")
import conftest  # noqa: E402,F401  (activates simulator mode)
import simlib  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402


class _Event:
  def __init__(self, enable_timing=False):
    self.t = None

  def record(self, stream=None):
    self.t = time.perf_counter()

  def elapsed_time(self, other):
    return (other.t - self.t) * 1e3


def _pin(t):
  p = simlib.host_tensor(tuple(t.shape), t.dtype, True)
  p.copy_(t)
  return p


torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 1
torch.cuda.set_device = lambda d: None
torch.cuda.current_stream = lambda d=None: types.SimpleNamespace(cuda_stream=0)
torch.cuda.synchronize = lambda d=None: None
torch.cuda.Event = _Event
torch.Tensor.pin_memory = _pin

import bench  # noqa: E402

tiny = ("squeezesegv2", "squeezesegv2", 32, 240, 3, 0.84, "hbm")
bench.WORKLOADS = dict(bench.WORKLOADS, tiny_ssv2=tiny, tiny_dn21=("darknet21", "darknet21", 16, 64, 2, 0.59, "mfma"))
bench.SECONDARY = (("tiny_dn21", 1, 1),)
bench.PARITY_WORKLOADS = ("tiny_ssv2",)
bench.REPEATS = 2
_np_load = np.load


def _load(path, *a, **k):      # c1_gpu_leg: 2 of the 32 real scans are enough here
  out = _np_load(path, *a, **k)
  if str(path).endswith("c1_sample_dataset_train_32x240.npz"):
    return {"raw": out["raw"][:2]}
  return out


np.load = _load
sys.argv = ["bench.py", "--workload", "tiny_ssv2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"]
# (argparse validates --workload against the WORKLOADS of import time)
bench.argparse.ArgumentParser.add_argument = (lambda orig: lambda self, *a, **k: orig(self, *a, **{kk: vv for kk, vv in k.items() if kk != "choices"}))(bench.argparse.ArgumentParser.add_argument)
bench.main()
