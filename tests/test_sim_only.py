"""Checks that only the functional simulator can make (PCLSEG_SIM=1 python -m pytest tests -m gpu; skipped on a real
device): what every launch of a forward pass actually asked for, against what pclseg_plan_ops says it will; results
that must not depend on the order in which waves and lanes run; and which kernel instantiations the pass executed.
tests/test_sim.py runs this file in simulator mode as part of the CPU suite."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import SIM

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not SIM, reason="simulator mode only (PCLSEG_SIM=1)")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NETS = [("squeezesegv2", "squeezesegv2kitti", 64, 256, 0.78), ("squeezesegv2", "squeezesegv2", 32, 240, 0.84),
        ("darknet21", "darknet21", 32, 128, 0.59), ("darknet53", "darknet53kitti", 16, 64, 0.78)]

_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import conftest   # activates simulator mode from PCLSEG_SIM
import pclsegmentation_amd as P
from pclsegmentation_amd import engine as E
from pclsegmentation_amd.utils.synthetic import synthetic_scans
model_name, cfg, h, w, pv, n, mb, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
mc, model = P.load_model_config(model_name, cfg, height=h, width=w)
model.init_weights(4321)
model.micro_batch = mb
raw = synthetic_scans(n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, pv, seed=5)
preds = np.empty((n, h, w), np.int32)
logits = np.empty((n, h, w, mc.NUM_CLASS), np.float32)
eng = model.engine(h, w)
open(sys.argv[9], "w").close()      # the trace starts here: finalize's packing kernels (if any) are not part of a pass
eng.forward_raw(raw, n, preds, None, logits, None, mem=E.MEM_HOST)
np.savez(out, preds=preds, logits=logits)
"""


def _forward_in_child(tmp_path, net, n, mb, tag, **env):
  """One forward pass in a fresh process (HIPSIM_* are read once per process) -> (outputs, launch trace)."""
  out, trace = str(tmp_path / ("out_%s.npz" % tag)), str(tmp_path / ("trace_%s.txt" % tag))
  e = dict(os.environ, HIPSIM_TRACE=trace, **env)
  subprocess.check_call([sys.executable, "-c", _CHILD % {"root": ROOT}, net[0], net[1], str(net[2]), str(net[3]), str(net[4]),
                         str(n), str(mb), out, trace], env=e, cwd=ROOT)
  rows = [l.rstrip("\n").split("\t") for l in open(trace)]
  return dict(np.load(out)), [(r[0], int(r[1]) * int(r[2]) * int(r[3]), int(r[4]), int(r[5])) for r in rows]


@pytest.mark.parametrize("net", NETS, ids=["%s_%dx%d" % (n[1], n[2], n[3]) for n in NETS])
def test_every_launch_asks_for_what_plan_ops_says(tmp_path, net):
  """pclseg_plan_ops derives LDS bytes / threads / blocks per scan from the graph with its own copy of the launchers'
  expressions (the three-lane residency analysis of DESIGN.md §12 rests on it).  The simulator sees the real launch
  arguments: one scan, one micro-batch — launch k of the pass must be row k of the plan."""
  import pclsegmentation_amd as P
  from pclsegmentation_amd import engine as E
  _, launches = _forward_in_child(tmp_path, net, 1, 1, "plan")
  mc, model = P.load_model_config(net[0], net[1], height=net[2], width=net[3])
  model.micro_batch = 1
  plan = E.plan_op_resources(model.engine_desc(net[2], net[3]))
  launches = [l for l in launches if "normalize_kernel" not in l[0]]          # pre-processing precedes the op list
  assert len(launches) == len(plan), (len(launches), len(plan), [l[0][:60] for l in launches], [p[0] for p in plan])
  for (kernel, blocks, threads, lds), (name, _macs, p_lds, p_threads, p_blocks) in zip(launches, plan):
    assert (blocks, threads, lds) == (p_blocks, p_threads, p_lds), \
      "%s: launched %s with %d blocks x %d threads, %d B LDS; plan_ops says %d x %d, %d B" % (
        name, kernel[:80], blocks, threads, lds, p_blocks, p_threads, p_lds)


@pytest.mark.parametrize("net", NETS[1:3], ids=["%s_%dx%d" % (n[1], n[2], n[3]) for n in NETS[1:3]])
def test_results_do_not_depend_on_the_order_waves_and_lanes_run_in(tmp_path, net):
  """A missing barrier (a wave reading LDS another wave has not yet written) makes the result depend on the schedule.
  Forward, reverse and two random orders of waves within a block and lanes within a wave: bit-identical outputs."""
  ref, _ = _forward_in_child(tmp_path, net, 3, 2, "fwd", HIPSIM_ORDER="fwd")
  for order in ("rev", "rand:1", "rand:2"):
    got, _ = _forward_in_child(tmp_path, net, 3, 2, order.replace(":", ""), HIPSIM_ORDER=order)
    assert np.array_equal(got["preds"], ref["preds"]) and np.array_equal(got["logits"].view(np.uint32), ref["logits"].view(np.uint32)), order


def test_bench_line_and_its_legs_on_the_simulator():
  """bench.py end to end (tests/bench_sim_child.py): main() with its five timed regions, host-boundary rows, exact-f32
  row, secondary workload, c1_gpu leg, parity_check rows and cpu_baseline, on shrunken workloads.  The JSON line must
  carry every field the driver and the judge read, c1_gpu must not have fallen into its error branch, and the parity
  rows must be green.  Timings printed here are meaningless."""
  import json
  r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_sim_child.py")], cwd=ROOT, capture_output=True, text=True,
                     timeout=1500, env=dict(os.environ))
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "spread", "repeats", "build", "host_boundary", "exact_f32",
              "secondary", "c1_gpu", "parity_check"):
    assert key in out, key
  assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0 and len(out["repeats"]["scans_per_s"]) == 2
  assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
  assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
  assert "error" not in out["c1_gpu"] and out["c1_gpu"]["page_locked_identical"] and out["c1_gpu"]["one_call_batch32"]["identical_to_batch1"]
  assert "predictions identical to the device-resident run: True" in out["host_boundary"]["note"]
  for row in out["parity_check"]:
    assert row["decided_identical"] and row["max_abs_logit_err_f16x3"] <= 1e-3 and row["max_abs_logit_err_f32"] <= 1e-3, row


_LAZY_CHILD = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import conftest, simlib
import torch
import pclsegmentation_amd as P
from pclsegmentation_amd import engine as E
from pclsegmentation_amd.utils.synthetic import synthetic_scans
lib = E.load_library()
assert lib.hipsim_streams_lazy() == 2
lib.hipsim_memcpy_async.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
lib.hipsim_memcpy_async.restype = None
mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
model.init_weights(4321)
model.micro_batch = 2
n, h, w = 7, 32, 240                      # 4 micro-batches over 3 lanes
raw = synthetic_scans(n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=21)
eng = model.engine(h, w)
ref = np.empty((n, h, w), np.int32)
eng.forward_raw(raw, n, ref, None, None, None, mem=E.MEM_HOST)
src = simlib.host_tensor((n, h, w, 5), torch.float32, True)      # page-locked: stays valid until the copy has run
src.copy_(torch.from_numpy(raw))
x = torch.zeros((n, h, w, 5), dtype=torch.float32)               # the "device" tensor, still empty
preds = torch.full((n, h, w), -5, dtype=torch.int32)
eng.set_stream(0)
lib.hipsim_memcpy_async(x.data_ptr(), src.data_ptr(), x.numel() * 4, None)     # queued on the caller's stream, not yet run
assert float(x.abs().sum()) == 0.0
eng.forward_raw(x, n, preds, None, None, None, mem=E.MEM_DEVICE)               # queued behind it
assert int((preds == -5).sum()) == preds.numel()                               # nothing has run
got = preds.cpu().numpy()                                                      # = wait for the caller's stream
print("IDENTICAL" if np.array_equal(got, ref) else "DIFFERENT %%d pixels" %% int((got != ref).sum()))
"""


def test_lanes_wait_for_what_the_callers_stream_has_queued(tmp_path):
  """HIPSIM_STREAMS=lazy: nothing runs before something waits for it.  The caller queues an upload of the scans on ITS
  stream and calls pclseg_forward_raw(MEM_DEVICE) right behind it; the engine's lane streams must wait for the caller's
  stream as it stood at the call (ev_in) and the caller's stream for the lanes at the end (join_lanes).  Waiting for
  the caller's stream alone must then produce the right predictions; a lane that did not wait would have read zeros."""
  r = subprocess.run([sys.executable, "-c", _LAZY_CHILD % {"root": ROOT}], cwd=ROOT, capture_output=True, text=True, timeout=900,
                     env=dict(os.environ, HIPSIM_STREAMS="lazy"))
  assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
  # control: with every hipStreamWaitEvent forgotten (fault injection in the simulator) the same program must come out wrong
  r = subprocess.run([sys.executable, "-c", _LAZY_CHILD % {"root": ROOT}], cwd=ROOT, capture_output=True, text=True, timeout=900,
                     env=dict(os.environ, HIPSIM_STREAMS="lazy", HIPSIM_DROP_WAITS="1"))
  assert "DIFFERENT" in r.stdout or r.returncode != 0, r.stdout[-1500:]


_FOOTPRINT_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import conftest
import pclsegmentation_amd as P
from pclsegmentation_amd import engine as E
from pclsegmentation_amd.utils.synthetic import synthetic_scans
mc, model = P.load_model_config("squeezesegv2", "squeezesegv2")
model.init_weights(4321)
model.micro_batch = 2
raw = synthetic_scans(2, 32, 240, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=3)
preds = np.empty((2, 32, 240), np.int32)
eng = model.engine(32, 240)
eng.forward_raw(raw, 2, preds, None, None, None, mem=E.MEM_HOST)
open(sys.argv[1], "w").close()
eng.forward_raw(raw, 2, preds, None, None, None, mem=E.MEM_HOST)
"""


def test_footprint_build_counts_known_tensor_sizes(tmp_path):
  """make -C sim traffic (scripts/sim_traffic.py): the lines a launch writes are its output tensor, the lines it reads its
  input tensor plus weights — known sizes for the first launches of SqueezeSegV2 at 2 x 32x240."""
  log = str(tmp_path / "traffic.txt")
  env = dict(os.environ, PCLSEG_SIM="traffic", HIPSIM_TRAFFIC_LOG=log, PCLSEG_LANES="1")
  subprocess.check_call([sys.executable, "-c", _FOOTPRINT_CHILD % {"root": ROOT}, log], cwd=ROOT, env=env)
  rows = [l.rstrip("\n").split("\t") for l in open(log)]
  assert len(rows) == 20                                     # pre-processing + 19 launches
  px = 2 * 32 * 240
  name, _, rd, wr, qrd, qwr = rows[0]
  assert "normalize_kernel" in name and int(wr) == px * 8 * 4 + px and int(rd) == px * 5 * 4      # [N,H,W,8] float32 + uint8 mask; the staged raw scans
  name, _, rd, wr, qrd, qwr = rows[1]                          # conv1: 3x3 stride 2, 8 -> 64 channels
  assert "conv_kernel" in name and int(wr) == px // 2 * 64 * 4
  assert px * 8 * 4 <= int(rd) <= px * 8 * 4 + 64 * 1024      # the padded input once, plus weights and bias
  assert int(qrd) > 2 * int(rd) and int(qwr) == int(wr)       # halos and cout groups re-request the input; every output byte is stored once
