"""CPU suite: a slice of the GPU tests, run on the functional simulator (sim/, tests/simlib.py).

The kernels and the host code of pclsegmentation_amd/csrc are compiled UNMODIFIED for x86-64 and executed wave by wave:
operator parity (both arithmetic modes, the wide 1x1 kernel included), whole networks against the oracle's golden
vectors, every intermediate tensor of SqueezeSegV2, the fused plan on ragged shapes, the host boundary, the C consumer,
launch geometry against pclseg_plan_ops, and independence of the wave / lane schedule.  This is a functional check — it
says nothing about speed and does not replace the run on an MI355X (`python -m pytest tests -m gpu` there); the whole
GPU suite runs here with `PCLSEG_SIM=1 python -m pytest tests -m gpu` (about ten minutes on 8 cores)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.sim      # ordered last by conftest.py; `-m "not gpu and not sim"` is the fast host suite


def _sim_pytest(args, timeout, **extra_env):
  env = dict(os.environ, PCLSEG_SIM="1", **extra_env)
  env.pop("PCLSEG_LIB", None)
  r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                     capture_output=True, text=True, timeout=timeout)
  tail = (r.stdout.strip().splitlines() or [""])[-1]
  m = re.search(r"(\d+) passed", tail)
  assert r.returncode == 0 and m and "failed" not in tail and "error" not in tail, r.stdout[-3000:] + r.stderr[-2000:]
  return int(m.group(1))


@pytest.fixture(scope="module")
def simulator():
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  import simlib
  return simlib.library("1")      # make -C sim (about 40 s when a source changed)


def test_operator_suite_on_the_simulator(simulator):
  assert _sim_pytest(["tests/test_gpu_ops.py"], 600) >= 67


def test_networks_host_boundary_and_c_consumer_on_the_simulator(simulator):
  k = ("(golden and f16x3 and (ssv2_32x240 or ssv2_real or darknet53kitti)) or (intermediate and squeezesegv2 and f16x3)"
       " or fully_fused or nan_pixel or c_consumer_forward")
  assert _sim_pytest(["tests/test_gpu_models.py", "tests/test_c_abi.py", "-k", k], 1500) >= 17


def test_candidate_kernels_on_the_simulator():
  """The candidate build (-DPCLSEG_CAND: tail pipelining, cam2 prefetch, two-pass slab, wide 1x1 kernel, KPIPE, GEOM 2,
  exact-mode epilogues — none of them run on an MI355X yet, none of them in the shipped library) must keep computing the
  oracle's values while it waits for its hardware A/B: the wide 1x1 operator shapes, SqueezeSegV2 and Darknet-53
  golden vectors, the fused plan on ragged shapes."""
  k = "wide_1x1 or (golden and f16x3 and (ssv2_32x240 or darknet53kitti)) or fully_fused"
  env = dict(os.environ, PCLSEG_SIM="cand")
  env.pop("PCLSEG_LIB", None)
  r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "tests/test_gpu_ops.py",
                      "tests/test_gpu_models.py", "-k", k], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
  tail = (r.stdout.strip().splitlines() or [""])[-1]
  m = re.search(r"(\d+) passed", tail)
  assert r.returncode == 0 and m and int(m.group(1)) >= 11 and "failed" not in tail, r.stdout[-3000:] + r.stderr[-2000:]


def test_launch_geometry_and_schedule_independence_on_the_simulator(simulator):
  k = "(plan_ops and (squeezesegv2_32x240 or darknet53kitti)) or (order and squeezesegv2_32x240)"
  assert _sim_pytest(["tests/test_sim_only.py", "-k", k], 900) >= 3


def test_stream_dependencies_with_lazy_internal_streams(simulator):
  """HIPSIM_STREAMS=internal: the caller's stream runs eagerly, every stream the engine creates (lanes, H2D, D2H) runs an
  operation only when something the caller's stream was made to wait for needs it.  A lane the engine forgot to join, a
  staging slab reused before its copy was waited for, an upload a kernel does not depend on: each would leave outputs
  stale here, every time, instead of once in a while on the device.  Multi-lane micro-batching, the three host
  boundaries (pinned, pageable, enqueue-only), the range-guard repair of queued calls and the device-tensor call
  surface must come out the same as in the eager schedule."""
  k = "micro_batch or host_boundary or range_fallback or call_surface"
  assert _sim_pytest(["tests/test_gpu_models.py", "-k", k], 1500, HIPSIM_STREAMS="internal") >= 4
  # control: with every hipStreamWaitEvent forgotten (fault injection in the simulator) the multi-lane test must FAIL
  env = dict(os.environ, PCLSEG_SIM="1", HIPSIM_STREAMS="internal", HIPSIM_DROP_WAITS="1")
  r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "tests/test_gpu_models.py", "-k", "micro_batch"],
                     cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode != 0 and "1 failed" in r.stdout, r.stdout[-1500:]
  # and the caller's-stream dependency in the fully lazy schedule (with its own control)
  assert _sim_pytest(["tests/test_sim_only.py", "-k", "lanes_wait"], 900) == 1


def test_bench_py_end_to_end_on_the_simulator(simulator):
  """bench.py's own code path (main, secondary rows, c1_gpu, parity_check, cpu_baseline) on shrunken workloads: the
  line it prints has every field of the contract.  Its timings mean nothing here."""
  assert _sim_pytest(["tests/test_sim_only.py", "-k", "bench_line"], 1500) == 1


def test_race_detector_controls_and_squeezesegv2():
  """sim/race_driver (ThreadSanitizer, one TSan fiber per wave; barriers and launch boundaries are the only
  happens-before edges): its controls — two seeded races it must report (LDS across two waves without the barrier, a
  global word stored by 64 blocks), three clean patterns it must not, the stream model's un-joined producer — and
  then SqueezeSegV2 forward passes in both arithmetic modes plus the projection / confusion-matrix operators with no
  report.  (All three networks: `sim/_build/race_driver`; full benchmark sizes: `race_driver full`.)"""
  subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sim"), "race"])
  exe = os.path.join(ROOT, "sim", "_build", "race_driver")
  env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=0")
  r = subprocess.run([exe, "selftest"], capture_output=True, text=True, timeout=600, env=env)
  rows = [l for l in r.stdout.splitlines() if l.startswith("selftest:")]
  assert r.returncode == 0 and len(rows) == 6 and all("as it must be" in l for l in rows), r.stdout + r.stderr[-2000:]
  assert sum("REPORTED" in l for l in rows) == 2      # LDS across two waves without the barrier; one global word stored by 64 blocks
  r = subprocess.run([exe, "ssv2", "ops"], capture_output=True, text=True, timeout=1500, env=env)
  assert r.returncode == 0 and "0 ThreadSanitizer report(s)" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
