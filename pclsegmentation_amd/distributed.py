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


def init_process_group(backend=None):
  """Initialise torch.distributed from MASTER_ADDR/MASTER_PORT/RANK/WORLD_SIZE.
  backend defaults to nccl (= RCCL on ROCm) when a GPU is visible, else gloo."""
  rank, local_rank, world = env_world()
  if world > 1 and not dist.is_initialized():
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
  if not dist.is_initialized() or dist.get_world_size() == 1:
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
  Darknet-53: one 216 MB collective instead of eight host-side fold + repack passes of 53 M parameters."""
  from . import engine as _engine
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return model.engine(height, width, flags)
  on_gpu = dist.get_backend() == "nccl"
  if on_gpu and device is None:
    device = torch.device("cuda", torch.cuda.current_device())
  if not on_gpu:
    device = torch.device("cpu")     # gloo moves host tensors (ranks sharing one GPU in tests)
  is_src = dist.get_rank() == src
  eng = model.engine(height, width, flags) if is_src else _engine.Engine(model.engine_desc(height, width, flags))
  blob = torch.empty(eng.packed_size(), dtype=torch.uint8, device=device)
  if is_src:
    eng.export_packed(blob)
  # The library copies on the legacy default stream, the collective runs on torch's own (non-blocking) RCCL
  # stream: neither is ordered against the other, so fence on both sides of the broadcast (start-up only).
  if on_gpu:
    torch.cuda.synchronize(device)
  dist.broadcast(blob, src=src)
  if on_gpu:
    torch.cuda.synchronize(device)
  if not is_src:
    eng.import_packed(blob)
    model.adopt_engine(eng, height, width, flags)
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
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return local_preds
  world, rank = dist.get_world_size(), dist.get_rank()
  counts = [shard_range(n_total, r, world) for r in range(world)]
  max_n = max(hi - lo for lo, hi in counts)
  pad = torch.zeros((max_n,) + tuple(local_preds.shape[1:]), dtype=local_preds.dtype,
                    device=local_preds.device)
  pad[:local_preds.shape[0]] = local_preds
  bufs = [torch.empty_like(pad) for _ in range(world)]
  dist.all_gather(bufs, pad)
  if rank != dst:
    return None
  return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, counts)], dim=0)
