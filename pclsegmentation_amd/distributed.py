"""Multi-GPU sharding: one process per GPU, scans split by contiguous ranges, weights
broadcast once over RCCL/xGMI.  The reference has no distributed code at all
(it only sets memory growth on physical_devices[0], inference.py:116-118),
so this is a new capability built for the path's natural parallelism: every scan is an
independent forward pass (BatchNorm in inference mode has no cross-sample statistics),
hence NO collective in steady state — only the start-up weight broadcast and, optionally,
a gather of int32 predictions.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
  """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
  return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
          int(os.environ.get("WORLD_SIZE", "1")))


def force_collectives():
  """PCLSEG_FORCE_COLLECTIVES=1 (test aid): a ONE-rank job still initialises the process group and runs
  every collective of the multi-GPU path — the status / packed-parameter broadcasts of broadcast_engine, the
  import of the broadcast blob into a fresh handle, bench.py's all_reduce of the timings — so the RCCL code
  path executes on a box with a single MI355X (tests/test_gpu_distributed.py)."""
  return os.environ.get("PCLSEG_FORCE_COLLECTIVES") == "1"


def collectives_active():
  """True when this job runs the multi-GPU code path: more than one rank, or a forced one-rank group."""
  return dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives())


def init_process_group(backend=None):
  """Initialise torch.distributed from MASTER_ADDR/MASTER_PORT/RANK/WORLD_SIZE.
  backend defaults to nccl (= RCCL on ROCm) when a GPU is visible, else gloo."""
  rank, local_rank, world = env_world()
  if (world > 1 or (force_collectives() and "MASTER_PORT" in os.environ)) and not dist.is_initialized():
    bind_rank(local_rank)            # (sysfs + sched_setaffinity only; nothing below has touched HIP yet)
    if backend is None:   # PCLSEG_DIST_BACKEND=gloo: test aid (several ranks sharing one GPU)
      backend = os.environ.get("PCLSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
      torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
  return rank, local_rank, world


def shard_range(n, rank, world):
  """Contiguous split of n scans: rank r owns [lo, hi); sizes differ by at most one."""
  base, rem = divmod(n, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def pack_weights(spec, weights):
  """Flatten a weight set into one float32 vector in spec order."""
  return np.concatenate([np.asarray(weights[w.path], np.float32).ravel() for w in spec])


def unpack_weights(spec, flat):
  out, off = {}, 0
  for w in spec:
    k = int(np.prod(w.shape))
    out[w.path] = np.asarray(flat[off:off + k], np.float32).reshape(w.shape)
    off += k
  if off != len(flat):
    raise ValueError("weight blob has %d scalars, spec needs %d" % (len(flat), off))
  return out


def broadcast_weights(spec, weights, src=0, device=None):
  """One broadcast of the packed weight blob (3.75 MB SqueezeSegV2 ... 212 MB Darknet-53).
  ``weights`` is only read on rank ``src``; every rank returns the full dict."""
  if not collectives_active():
    return weights
  n = sum(int(np.prod(w.shape)) for w in spec)
  if dist.get_backend() != "nccl":
    device = torch.device("cpu")     # gloo moves host tensors (CPU tests, ranks sharing one GPU)
  elif device is None:
    device = torch.device("cuda", torch.cuda.current_device())
  if dist.get_rank() == src:
    blob = torch.from_numpy(pack_weights(spec, weights)).to(device)
  else:
    blob = torch.empty(n, dtype=torch.float32, device=device)
  dist.broadcast(blob, src=src)
  return unpack_weights(spec, blob.cpu().numpy())


def broadcast_engine(model, height, width, flags=0, src=0, device=None):
  """Every rank gets a ready engine of ``model`` at height x width; only rank ``src`` needs the
  weights.  Rank ``src`` folds BatchNorm and packs the MFMA fragments once (pclseg_finalize), exports
  the packed device arrays into one buffer (pclseg_export_packed), ONE broadcast moves it — over
  RCCL/xGMI device to device — and the other ranks import it (pclseg_import_packed: a device copy).
  Darknet-53: one 216 MB collective instead of eight host-side fold + repack passes of 53 M parameters.

  Two small collectives guard the big one, so that no rank is ever left waiting in the blob's collective
  until the RCCL / gloo timeout: a 16-byte status broadcast (rank ``src`` could not build its engine: no
  weights bound, pclseg_finalize refusing non-finite folded weights, out of memory), then — after every
  OTHER rank has created its receiving engine and compared its packed size — a MIN all-reduce of
  ``rank if failed else world`` (out of memory with several ranks on one GPU, different library builds or
  fusion switches between ranks).  If either reports a failure EVERY rank raises, naming the failing rank.

  With PCLSEG_FORCE_COLLECTIVES=1 a one-rank group runs the same sequence and returns the engine that
  IMPORTED the broadcast blob (the source engine is dropped), so a single-GPU box executes the transport."""
  from . import engine as _engine
  if not collectives_active():
    return model.engine(height, width, flags)
  on_gpu = dist.get_backend() == "nccl"
  if on_gpu:
    want = torch.device("cuda", int(model.device))
    if device is not None and torch.device(device) != want:
      raise ValueError("broadcast_engine: model.device is %s but the blob was asked for on %s" % (want, device))
    device = want
    torch.cuda.set_device(device)   # RCCL binds a rank's communicator to the current device
  else:
    device = torch.device("cpu")     # gloo moves host tensors (ranks sharing one GPU in tests)
  is_src = dist.get_rank() == src
  loopback = dist.get_world_size() == 1
  eng, err = None, None
  status = torch.zeros(2, dtype=torch.int64, device=device)      # [ok, packed bytes]
  if is_src:
    try:
      eng = model.engine(height, width, flags)
      status[0], status[1] = 1, eng.packed_size()
    except Exception as e:      # reported to every rank below, then re-raised here
      err = e
  dist.broadcast(status, src=src)
  ok, nbytes = int(status[0].item()), int(status[1].item())
  if not ok:
    if err is not None:
      raise err
    raise RuntimeError("broadcast_engine: rank %d could not build the engine (see its log)" % src)
  dst = None
  world, rank = dist.get_world_size(), dist.get_rank()
  if not is_src or loopback:
    try:
      dst = _engine.Engine(model.engine_desc(height, width, flags))
      if dst.packed_size() != nbytes:
        raise RuntimeError("broadcast_engine: rank %d packs %d bytes, this rank's plan needs %d "
                           "(different library builds or fusion switches between ranks?)" % (src, nbytes, dst.packed_size()))
    except Exception as e:      # every rank learns of it below, then it is re-raised here
      err = e
  first_bad = torch.full((1,), world if err is None else rank, dtype=torch.int64, device=device)
  dist.all_reduce(first_bad, op=dist.ReduceOp.MIN)
  first_bad = int(first_bad.item())
  if first_bad < world:
    if dst is not None:
      dst.close()
    if err is not None:
      raise err
    raise RuntimeError("broadcast_engine: rank %d could not create its receiving engine (see its log)" % first_bad)
  blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
  if is_src:
    eng.export_packed(blob)
  # The library copies on the legacy default stream, the collective runs on torch's own (non-blocking) RCCL
  # stream: neither is ordered against the other, so fence on both sides of the broadcast (start-up only).
  if on_gpu:
    torch.cuda.synchronize(device)
  dist.broadcast(blob, src=src)
  if on_gpu:
    torch.cuda.synchronize(device)
  if not is_src or loopback:
    dst.import_packed(blob)
    if loopback:
      model._drop_engines()          # the returned engine is the one that came through the collective
    model.adopt_engine(dst, height, width, flags)
    eng = dst
  return eng


def bind_rank(local_rank):
  """Pin this rank process to the CPU cores next to its GPU.  Must run BEFORE the process touches HIP
  (it only reads sysfs and calls sched_setaffinity): KFD topology node order is the HIP device order,
  a node's PCI address comes from its `domain` / `location_id` properties, and
  /sys/bus/pci/devices/<bdf>/local_cpulist names the NUMA-local cores.  A speed hint only: any
  failure — or a *_VISIBLE_DEVICES variable that re-maps device numbers — leaves the affinity alone."""
  import glob
  if any(os.environ.get(v) for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
    return None
  try:
    gpus = []
    nodes = sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p)))
    for node in nodes:
      props = dict(l.split()[:2] for l in open(os.path.join(node, "properties")).read().splitlines()
                   if len(l.split()) >= 2)
      if int(props.get("simd_count", "0")) > 0:
        loc = int(props["location_id"])
        gpus.append("%04x:%02x:%02x.%x" % (int(props.get("domain", "0")), (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7))
    if not gpus:
      return None
    bdf = gpus[local_rank % len(gpus)]
    cpus = set()
    for part in open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip().split(","):
      lo, _, hi = part.partition("-")
      cpus.update(range(int(lo), int(hi or lo) + 1))
    allowed = cpus & os.sched_getaffinity(0)
    if allowed:
      os.sched_setaffinity(0, allowed)
      return sorted(allowed)
  except (OSError, ValueError, KeyError):
    pass
  return None


def gather_predictions(local_preds, n_total, dst=0):
  """Optional: collect every rank's int32 predictions [n_local,H,W] on rank ``dst`` in scan
  order (ranks may hold different counts)."""
  if not collectives_active():
    return local_preds
  world, rank = dist.get_world_size(), dist.get_rank()
  counts = [shard_range(n_total, r, world) for r in range(world)]
  max_n = max(hi - lo for lo, hi in counts)
  pad = torch.zeros((max_n,) + tuple(local_preds.shape[1:]), dtype=local_preds.dtype,
                    device=local_preds.device)
  pad[:local_preds.shape[0]] = local_preds
  # gather, not all_gather: only `dst` receives (and allocates) the other ranks' 16.8 MB shards (C4: 32 scans of
  # 64x2048 int32 per GPU) — 1/world of the xGMI traffic of handing every shard to every rank
  bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
  dist.gather(pad, gather_list=bufs, dst=dst)
  if rank != dst:
    return None
  return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, counts)], dim=0)
