"""The N > 1 path with real engine ranks on the GPU.  The test box has ONE MI355X, so the two ranks
share it and rendezvous over gloo (PCLSEG_DIST_BACKEND=gloo: RCCL cannot place two ranks on one
device); everything else is the production path — fresh rank processes started by
torch.distributed.run, one weight broadcast from rank 0, contiguous scan shards, no data-path
collective, optional gather.  Each rank's result must be bit-identical to the single-process result
for its shard (scans are independent: nets/SegmentationNetwork.py:133-136, BatchNorm in inference mode)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import pclsegmentation_amd as P
from pclsegmentation_amd import distributed as D
from pclsegmentation_amd import engine as E
from pclsegmentation_amd.utils.synthetic import synthetic_scan_range, synthetic_scans

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _rank_env():
  env = dict(os.environ)
  env.update(PCLSEG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
    env.pop(k, None)
  return env


@pytest.mark.needs_hip
@pytest.mark.parametrize("world,n", [(2, 7), (8, 19)])   # 7 scans over 2 ranks: [0,4) [4,7); 19 over 8: ragged 3,3,3,2,...
def test_engine_ranks_reproduce_the_single_process_result(cuda, tmp_path, world, n):
  h, w = 32, 240
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
         os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(n), str(h), str(w)]
  r = subprocess.run(cmd, env=_rank_env(), capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  # single-process reference on this process's engine
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2", height=h, width=w)
  model.init_weights(4321)
  raw = synthetic_scan_range(0, n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=99)
  eng = model.engine(h, w)
  preds = np.empty((n, h, w), np.int32)
  logits = np.empty((n, h, w, mc.NUM_CLASS), np.float32)
  eng.forward_raw(raw, n, preds, None, logits, None, mem=E.MEM_HOST)
  seen = np.zeros(n, bool)
  for rank in range(world):
    g = np.load(str(tmp_path / ("rank%d.npz" % rank)))
    lo, hi = int(g["lo"]), int(g["hi"])
    assert (lo, hi) == D.shard_range(n, rank, world)
    assert np.array_equal(g["preds"], preds[lo:hi]) and np.array_equal(g["logits"], logits[lo:hi])
    seen[lo:hi] = True
  assert seen.all()
  assert np.array_equal(np.load(str(tmp_path / "gathered.npy")), preds)   # optional gather, scan order


def _torchrun(world, script_args, env, timeout=900):
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
  return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.needs_hip
def test_one_rank_nccl_runs_the_rccl_transport(cuda, tmp_path):
  """The RCCL code path on the one GPU this box has: ONE rank under torch.distributed.run with the nccl
  backend and PCLSEG_FORCE_COLLECTIVES=1 — init_process_group's nccl branch, broadcast_engine's status
  broadcast, export -> dist.broadcast(device uint8 blob) -> pclseg_import_packed(MEM_DEVICE) into a fresh
  handle, a forward pass on THAT handle, all_gather of device predictions and an all_reduce of a device
  tensor.  Outputs must be bit-identical to the plain engine's.  What stays unexercised is the xGMI wire."""
  h, w, n = 32, 240, 5
  env = _rank_env()
  env.pop("PCLSEG_DIST_BACKEND")
  env["PCLSEG_FORCE_COLLECTIVES"] = "1"
  r = _torchrun(1, [os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(n), str(h), str(w)], env, 600)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  g = np.load(str(tmp_path / "rank0.npz"))
  assert str(g["backend"]) == "nccl" and int(g["world"]) == 1 and bool(g["engine_came_through_collective"])
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2", height=h, width=w)
  model.init_weights(4321)
  raw = synthetic_scan_range(0, n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=99)
  preds = np.empty((n, h, w), np.int32)
  logits = np.empty((n, h, w, mc.NUM_CLASS), np.float32)
  model.engine(h, w).forward_raw(raw, n, preds, None, logits, None, mem=E.MEM_HOST)
  assert np.array_equal(g["preds"], preds) and np.array_equal(g["logits"], logits)
  assert np.array_equal(np.load(str(tmp_path / "gathered.npy")), preds)
  model._drop_engines()


def _device_count():
  import torch
  return torch.cuda.device_count()      # (does not initialise HIP on this image)


@pytest.mark.needs_hip
@pytest.mark.first_hw_run
@pytest.mark.skipif(_device_count() < 2, reason="needs two MI355X: arms itself on the first multi-GPU box")
def test_two_rank_nccl_over_xgmi(cuda, tmp_path):
  """The production N > 1 path on REAL devices: one rank per GPU, nccl (= RCCL) backend, NO forced collectives and no
  gloo stand-in — broadcast_engine moves the packed parameters device to device over xGMI, each rank runs its
  contiguous shard with no data-path collective, gather_predictions runs on the device.  Every rank's predictions
  and logits must be bit-identical to the single-process result for its shard (the reference has no counterpart:
  inference.py:116-118 is one process).  Skipped on a one-GPU box."""
  world = min(_device_count(), 8)
  h, w, n = 32, 240, 2 * world + 3          # ragged shards
  env = _rank_env()
  env.pop("PCLSEG_DIST_BACKEND")            # -> nccl
  env.pop("PCLSEG_FORCE_COLLECTIVES", None)
  r = _torchrun(world, [os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(n), str(h), str(w)], env, 900)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2", height=h, width=w)
  model.init_weights(4321)
  raw = synthetic_scan_range(0, n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=99)
  preds = np.empty((n, h, w), np.int32)
  logits = np.empty((n, h, w, mc.NUM_CLASS), np.float32)
  model.engine(h, w).forward_raw(raw, n, preds, None, logits, None, mem=E.MEM_HOST)
  seen = np.zeros(n, bool)
  for rank in range(world):
    g = np.load(str(tmp_path / ("rank%d.npz" % rank)))
    lo, hi = int(g["lo"]), int(g["hi"])
    assert str(g["backend"]) == "nccl" and int(g["world"]) == world and (lo, hi) == D.shard_range(n, rank, world)
    assert bool(g["engine_came_through_collective"]) == (rank != 0)
    assert np.array_equal(g["preds"], preds[lo:hi]) and np.array_equal(g["logits"], logits[lo:hi]), "rank %d" % rank
    seen[lo:hi] = True
  assert seen.all()
  assert np.array_equal(np.load(str(tmp_path / "gathered.npy")), preds)      # all_gather of device tensors, scan order
  model._drop_engines()


@pytest.mark.needs_hip
def test_bench_one_rank_nccl_reduces_its_timings_on_the_device(cuda):
  """bench.py as the driver launches it for N > 1 (under torch.distributed.run), here with one forced rank:
  nccl process group, broadcast_engine through RCCL, barrier-fenced timed region, all_reduce(MAX) of the
  timing tensor ON THE DEVICE (bench.py's multi-GPU branch)."""
  env = _rank_env()
  env.pop("PCLSEG_DIST_BACKEND")
  env["PCLSEG_FORCE_COLLECTIVES"] = "1"
  r = _torchrun(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                    "--workload", "ssv2_32x240", "--cpu-seconds", "0", "--no-secondary"], env)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
  assert out["config"]["collectives"] == {"backend": "nccl", "world": 1, "timing_reduced_on": "device"}
  assert out["value"] > 0 and out["n_gpus"] == 1


@pytest.mark.needs_hip
def test_c4_full_size_256_scans_over_8_ranks(cuda, tmp_path):
  """BASELINE configs[3] (C4) functionally: 256 scans of 64x2048 (SqueezeSegV2, 20 classes) sharded over 8
  fresh rank processes — 32 scans each, one packed-parameter broadcast, no data-path collective.  The ranks
  share this box's one MI355X and rendezvous over gloo; every rank's predictions (and the logits of its
  first scan) must equal the single-process result for shard_range(256, r, 8)."""
  h, w, n, world = 64, 2048, 256, 8
  r = _torchrun(world, [os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(n), str(h), str(w),
                        "squeezesegv2kitti"], _rank_env(), 1500)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2kitti", height=h, width=w)
  model.init_weights(4321)
  eng = model.engine(h, w)
  covered = 0
  for rank in range(world):
    g = np.load(str(tmp_path / ("rank%d.npz" % rank)))
    lo, hi = int(g["lo"]), int(g["hi"])
    assert (lo, hi) == D.shard_range(n, rank, world) == (32 * rank, 32 * rank + 32)
    raw = synthetic_scan_range(lo, hi, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.78, seed=99)
    preds = np.empty((hi - lo, h, w), np.int32)
    eng.forward_raw(raw, hi - lo, preds, None, None, None, mem=E.MEM_HOST)
    assert np.array_equal(g["preds"], preds), "rank %d" % rank
    first, logits = np.empty((1, h, w), np.int32), np.empty((1, h, w, mc.NUM_CLASS), np.float32)
    eng.forward_raw(raw, 1, first, None, logits, None, mem=E.MEM_HOST)
    assert np.array_equal(g["logits"], logits) and np.array_equal(first[0], preds[0])
    assert len(np.unique(preds)) > 3          # not a degenerate map
    covered += hi - lo
  assert covered == n
  model._drop_engines()


@pytest.mark.needs_hip
def test_bench_c4_strong_scaling_full_size(cuda):
  """`bench.py --gpus 8 --scaling strong --global-batch 256` at the headline shape: the driver's C4 command
  line, with the 8 ranks sharing one GPU (functional check of the launch, sharding and reduction)."""
  env = _rank_env()
  env.pop("PCLSEG_DIST_BACKEND")
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
         "--cpu-seconds", "0", "--scaling", "strong", "--global-batch", "256"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
  assert out["n_gpus"] == 8 and out["scaling"] == "strong"
  assert out["config"]["workload"] == "ssv2_64x2048" and out["config"]["shape"] == [64, 2048]
  assert out["config"]["global_batch"] == 256 and out["config"]["batch_per_gpu"] == 32
  assert out["config"]["collectives"]["backend"] == "gloo" and out["value"] > 0


@pytest.mark.needs_hip
def test_bench_launches_its_own_ranks(cuda):
  """`python bench.py --gpus 2` outside torchrun starts its rank processes itself (the driver's
  launch line for N > 1 without torch.distributed.run) and prints ONE JSON line for the job."""
  env = _rank_env()
  env.pop("PCLSEG_DIST_BACKEND")      # bench.py chooses gloo itself when ranks outnumber GPUs
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
         "--workload", "ssv2_32x240", "--cpu-seconds", "0"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout[-2000:]
  out = json.loads(lines[0])
  assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"]
  assert out["value"] > 0 and out["scaling"] == "weak" and "roofline" in out


@pytest.mark.needs_hip
def test_bench_strong_scaling_mode(cuda):
  """`--scaling strong`: a FIXED global batch sharded over the ranks by contiguous ranges (SURVEY.md §8(e),
  BASELINE configs[3] is 256 scans over 8 GPUs); here 10 scans over 4 ranks sharing the one GPU."""
  env = _rank_env()
  env.pop("PCLSEG_DIST_BACKEND")
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1",
         "--workload", "ssv2_32x240", "--cpu-seconds", "0", "--scaling", "strong", "--global-batch", "10"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
  assert out["n_gpus"] == 4 and out["scaling"] == "strong" and out["config"]["global_batch"] == 10
  assert out["config"]["batch_per_gpu"] == 3       # rank 0's share of 10 over 4: 3,3,2,2
  assert abs(out["value"] - 10 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-2


def test_packed_parameters_round_trip(cuda):
  """pclseg_export_packed -> pclseg_import_packed (host and device buffers): the importing handle, which
  never saw a Keras tensor, produces bit-identical outputs; a blob made for another desc is refused."""
  import torch
  h, w = 32, 240
  mc, model = P.load_model_config("darknet21", "darknet21", height=h, width=w)
  model.init_weights(4321)
  raw = synthetic_scans(3, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=5)
  for flags in (0, E.FLAG_EXACT_F32, E.FLAG_RANGE_FALLBACK):
    src = model.engine(h, w, flags)
    want_p, want_l = np.empty((3, h, w), np.int32), np.empty((3, h, w, mc.NUM_CLASS), np.float32)
    src.forward_raw(raw, 3, want_p, None, want_l, None, mem=E.MEM_HOST)
    nbytes = src.packed_size()
    for blob in (np.empty(nbytes, np.uint8), torch.empty(nbytes, dtype=torch.uint8, device=cuda)):
      src.export_packed(blob)
      dst = E.Engine(model.engine_desc(h, w, flags))
      dst.import_packed(blob)
      p, l = np.empty_like(want_p), np.empty_like(want_l)
      dst.forward_raw(raw, 3, p, None, l, None, mem=E.MEM_HOST)
      assert np.array_equal(p, want_p) and np.array_equal(l, want_l)
      with pytest.raises(RuntimeError):
        dst.import_packed(blob)                      # already finalized
      dst.close()
    other = E.Engine(model.engine_desc(h, 256, flags))
    with pytest.raises(ValueError, match="does not fit"):
      other.import_packed(blob)
    other.close()
    with pytest.raises(ValueError):
      src.export_packed(np.empty(nbytes - 1, np.uint8))
  model._drop_engines()
