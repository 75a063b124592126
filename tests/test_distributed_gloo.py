"""The N > 1 path on CPU: two processes over gloo — contiguous scan sharding with no data-path
collective, one weight broadcast, optional prediction gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pclsegmentation_amd import distributed as D
from pclsegmentation_amd.nets.spec import squeezesegv2_spec
from pclsegmentation_amd.nets.weights import synthetic_weights


def test_shard_range_partitions():
  for n in (0, 1, 5, 32, 255, 256):
    for world in (1, 2, 3, 8):
      spans = [D.shard_range(n, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1
  assert [D.shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]


def test_pack_unpack_round_trip():
  spec = squeezesegv2_spec(11)
  w = synthetic_weights(spec, 3)
  flat = D.pack_weights(spec, w)
  assert flat.dtype == np.float32 and flat.size == 931887
  back = D.unpack_weights(spec, flat)
  assert all(np.array_equal(w[k], back[k]) for k in w)


def _worker(rank, world, port, q):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                    LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
  r, _, ws = D.init_process_group(backend="gloo")
  assert (r, ws) == (rank, world)
  spec = squeezesegv2_spec(11)
  weights = synthetic_weights(spec, 4321) if rank == 0 else None
  weights = D.broadcast_weights(spec, weights, src=0, device=torch.device("cpu"))
  digest = float(sum(np.float64(v).sum() for v in weights.values()))
  lo, hi = D.shard_range(5, rank, world)
  local = torch.full((hi - lo, 2, 3), rank, dtype=torch.int32)      # "predictions" of my scans
  local += torch.arange(lo, hi, dtype=torch.int32).view(-1, 1, 1) * 10
  full = D.gather_predictions(local, 5, dst=0)
  q.put((rank, digest, (lo, hi), None if full is None else full[:, 0, 0].tolist()))
  dist.barrier()
  dist.destroy_process_group()


def test_two_process_broadcast_and_sharding():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted(q.get(timeout=120) for _ in range(2))
  for p in procs:
    p.join(60)
    assert p.exitcode == 0
  want = float(sum(np.float64(v).sum() for v in synthetic_weights(squeezesegv2_spec(11), 4321).values()))
  assert res[0][1] == want and res[1][1] == want          # rank 1 received rank 0's weights
  assert res[0][2] == (0, 3) and res[1][2] == (3, 5)      # disjoint, covering
  assert res[0][3] == [0, 10, 20, 31, 41] and res[1][3] is None


def _failing_src_worker(rank, world, port, q):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                    LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
  import pclsegmentation_amd as P
  D.init_process_group(backend="gloo")
  _, model = P.load_model_config("squeezesegv2", "squeezesegv2")     # NO weights bound on any rank
  try:
    D.broadcast_engine(model, 32, 240, src=0)
    q.put((rank, "returned"))
  except RuntimeError as e:
    q.put((rank, "RuntimeError: %s" % e))
  dist.barrier()
  dist.destroy_process_group()


def test_broadcast_engine_failure_on_the_source_rank_raises_everywhere():
  """Rank 0 cannot build its engine (no weights): the status broadcast that precedes the blob makes rank 1
  raise too, instead of waiting in a collective that rank 0 never joins."""
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_failing_src_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = dict(q.get(timeout=120) for _ in range(2))
  for p in procs:
    p.join(60)
    assert p.exitcode == 0
  assert "no weights" in res[0] and res[0].startswith("RuntimeError")
  assert "rank 0 could not build the engine" in res[1]


class _FakeEngine:
  """Stands in for engine.Engine on a box without a GPU: only the control flow of broadcast_engine is under test."""
  def __init__(self, desc=None, nbytes=64):
    self.nbytes = nbytes
  def packed_size(self):
    return self.nbytes
  def export_packed(self, blob):
    blob.fill_(7)
  def import_packed(self, blob):
    self.got = int(blob.sum())
  def close(self):
    pass


def _failing_dst_worker(rank, world, port, q, mode):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                    LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
  import pclsegmentation_amd as P
  from pclsegmentation_amd import engine as E
  D.init_process_group(backend="gloo")
  _, model = P.load_model_config("squeezesegv2", "squeezesegv2")
  model.engine = lambda h, w, flags=0: _FakeEngine()
  model.engine_desc = lambda h, w, flags=0: None
  model.adopt_engine = lambda eng, h, w, flags=0: None
  if mode == "create_fails":
    def boom(desc):
      raise MemoryError("pclseg: hipMalloc(activation arena): out of memory")
    E.Engine = boom if rank == 2 else _FakeEngine
  else:       # this rank's plan packs a different number of bytes
    E.Engine = (lambda desc: _FakeEngine(nbytes=80)) if rank == 1 else _FakeEngine
  try:
    D.broadcast_engine(model, 32, 240, src=0)
    q.put((rank, "returned"))
  except (RuntimeError, MemoryError) as e:
    q.put((rank, "%s: %s" % (type(e).__name__, e)))
  dist.barrier()
  dist.destroy_process_group()


def test_broadcast_engine_failure_on_a_receiving_rank_raises_everywhere():
  """A NON-source rank cannot create its receiving engine (out of memory) or packs a different size: the MIN
  all-reduce that follows the status broadcast makes every rank raise and name the failing rank — rank 0 is not
  left waiting in the blob's broadcast."""
  for mode, world, bad, text in (("create_fails", 3, 2, "out of memory"), ("size_differs", 2, 1, "this rank's plan needs 80")):
    with socket.socket() as s:
      s.bind(("127.0.0.1", 0))
      port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_dst_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
      p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
      p.join(60)
      assert p.exitcode == 0
    assert text in res[bad], res
    for r in range(world):
      if r != bad:
        assert "rank %d could not create its receiving engine" % bad in res[r], res


def _one_rank_worker(port, q):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                    PCLSEG_FORCE_COLLECTIVES="1")
  r, _, ws = D.init_process_group(backend="gloo")
  ok = dist.is_initialized() and D.collectives_active() and (r, ws) == (0, 1)
  spec = squeezesegv2_spec(11)
  w = D.broadcast_weights(spec, synthetic_weights(spec, 1), src=0)
  full = D.gather_predictions(torch.arange(6, dtype=torch.int32).view(3, 2, 1), 3, dst=0)
  q.put((ok, len(w), full[:, 0, 0].tolist()))
  dist.destroy_process_group()


def test_forced_one_rank_group_runs_the_collectives():
  """PCLSEG_FORCE_COLLECTIVES=1: a one-rank job initialises the group and goes through the collectives
  (the switch behind the one-rank RCCL test on the GPU box)."""
  assert not D.collectives_active()
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  p = ctx.Process(target=_one_rank_worker, args=(port, q))
  p.start()
  ok, nw, first = q.get(timeout=120)
  p.join(60)
  assert p.exitcode == 0 and ok and nw == len(squeezesegv2_spec(11)) and first == [0, 2, 4]


def test_rank_binding_and_gpu_census_never_touch_hip():
  """bench.py's launcher and distributed.bind_rank decide from sysfs / the environment only; both must
  work (and do nothing harmful) on a box without a GPU."""
  import importlib.util
  import sys as _sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  n = bench.visible_gpu_count()
  assert isinstance(n, int) and n >= 0
  before = os.sched_getaffinity(0)
  got = D.bind_rank(0)
  after = os.sched_getaffinity(0)
  assert got is None or (set(got) == after and after <= before)
  os.sched_setaffinity(0, before)
  os.environ["HIP_VISIBLE_DEVICES"] = "0"
  try:
    assert D.bind_rank(3) is None and os.sched_getaffinity(0) == before      # re-mapped device numbers: hands off
    assert bench.visible_gpu_count() <= 1
  finally:
    del os.environ["HIP_VISIBLE_DEVICES"]
