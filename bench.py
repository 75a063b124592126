#!/usr/bin/env python3
"""Benchmark of the hot path: LiDAR scans/s of the forward pass on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a torchrun environment: bench.py starts N fresh rank processes itself
  (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...) BEFORE
  anything in this process touches the GPU, and exits with their status; launched by the driver
  under torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

One step = pclseg_forward_raw over one batch of synthetic raw scans already resident in HBM
(normalise+mask -> network -> argmax -> masked int32 predictions, also left in HBM).  Default
workload = BASELINE.json configs[1]: SqueezeSegV2, 64x2048, 20 classes, batch 32 per GPU, random
(seeded) weights of that architecture.  Multi-GPU = weak scaling: every rank runs its own 32-scan
batch, weights broadcast once over RCCL, no collective inside the timed region.

Rank 0 prints ONE JSON line (see DESIGN.md §Measurement for the field definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TF = 157.3     # dense f32-input MFMA peak (= f32 vector peak)
F16_MFMA_PEAK_TF = 2500.0    # dense f16 MFMA peak; the split-f16 path spends 3 MFMA products per MAC

WORKLOADS = {
  # name: (model, config, H, W, batch per GPU, valid-pixel rate, roofline bound)
  "ssv2_64x2048": ("squeezesegv2", "squeezesegv2kitti", 64, 2048, 32, 0.78, "hbm"),
  "darknet53_64x2048": ("darknet53", "darknet53kitti", 64, 2048, 16, 0.78, "mfma"),
  "darknet21_32x1024": ("darknet21", "darknet21", 32, 1024, 64, 0.59, "mfma"),
  "ssv2_32x240": ("squeezesegv2", "squeezesegv2", 32, 240, 32, 0.84, "hbm"),
}
# the other single-GPU configurations of BASELINE.json timed in the secondary rows: (workload, steps, warm-up)
SECONDARY = (("darknet53_64x2048", 20, 4), ("darknet21_32x1024", 20, 4))
# workloads of the parity_check rows (one full-size scan each against the float64 oracle)
PARITY_WORKLOADS = ("ssv2_64x2048", "darknet53_64x2048", "darknet21_32x1024")


def _cpu_leg(net, mc, raw, threads, batch, budget_s, max_scans):
  """scans/s of the CPU stand-in at a given thread count and batch size (bounded sample)."""
  import torch
  from oracle import np_oracle as O
  torch.set_num_threads(threads)
  none_index = mc.CLASSES.index("None")

  def one_batch(i):
    chunk = raw[(i * batch) % raw.shape[0]:][:batch]
    lidar, mask = O.normalize_and_mask(chunk, mc.INPUT_MEAN, mc.INPUT_STD)
    net(lidar.astype(np.float32), mask, none_index)
    return chunk.shape[0]

  one_batch(0)  # warm-up (oneDNN primitive creation)
  done, i, t0 = 0, 0, time.perf_counter()
  while True:
    done += one_batch(i)
    i += 1
    el = time.perf_counter() - t0
    if el >= budget_s or done >= max_scans:
      break
  return round(done / el, 3), done


def cpu_baseline(model_name, mc, weights, h, w, pvalid, budget_s):
  """Time the CPU stand-in for the reference's TF2-CPU path on this box's host cores:
  oracle/torch_ref.py (PyTorch-CPU/oneDNN expression of the identical graph, identical weights).
  Bounded sample.  Legs: all usable cores at batch 2 (`value`), the reference's
  own loop shape B = 1 (inference.py:44-75) at the same thread count, one thread at B = 1, and —
  for the SqueezeSegV2 headline — config C1 on the reference's real 32x240 sample scans."""
  import torch
  from oracle.torch_ref import TorchNet
  from pclsegmentation_amd.utils.synthetic import synthetic_scans
  # oneDNN scales poorly past a few dozen threads on these small convolutions (256 threads ran
  # 20x slower than 8 on the GPU box), so the pool is capped; `cores` reports the threads used
  cores = min(os.cpu_count() or 1, 32)
  net = TorchNet(model_name, weights, num_layers=mc.get("NUM_LAYERS"),
                 output_stride=mc.get("OUTPUT_STRIDE", 16))
  raw = synthetic_scans(2, h, w, mc.INPUT_MEAN, mc.INPUT_STD, pvalid, seed=77)
  main, done = _cpu_leg(net, mc, raw, cores, 2, 0.5 * budget_s, 64)
  b1, done_b1 = _cpu_leg(net, mc, raw, cores, 1, 0.2 * budget_s, 32)
  t1, done_t1 = _cpu_leg(net, mc, raw, 1, 1, 0.3 * budget_s, 8)
  out = {"value": main, "unit": "scans/s", "cores": cores, "kind": "port",
         "sample": "%d synthetic %dx%d scans, batch 2, PyTorch-CPU (oneDNN) expression of the same "
                   "graph and weights (oracle/torch_ref.py); TF2 itself is not installable here"
                   % (done, h, w),
         "legs": {"batch1_%dthreads" % cores: {"value": b1, "scans": done_b1},
                  "batch1_1thread": {"value": t1, "scans": done_t1}}}
  real = os.path.join(ROOT, "tests", "golden", "c1_sample_dataset_train_32x240.npz")
  if model_name == "squeezesegv2" and os.path.exists(real):
    # BASELINE.json configs[0] (C1): SqueezeSegV2 / NC 11 on the 32 real scans of the reference's
    # dataset_samples/sample_dataset/train (committed as a fixture), the reference's B = 1 loop
    import pclsegmentation_amd as P
    from pclsegmentation_amd.nets.weights import synthetic_weights
    mc1, m1 = P.load_model_config("squeezesegv2", "squeezesegv2")
    net1 = TorchNet("squeezesegv2", synthetic_weights(m1.weight_spec(), 4321))
    raw1 = np.load(real)["raw"]
    c1, done_c1 = _cpu_leg(net1, mc1, raw1, cores, 1, 3.0, 32)   # one pass over the 32 real scans at most
    out["legs"]["c1_real_32x240_batch1_%dthreads" % cores] = {"value": c1, "scans": done_c1}
  torch.set_num_threads(cores)
  return out


def _timed(fn, stream, steps, warmup):
  import torch
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(stream)
  for _ in range(steps):
    fn()
  e1.record(stream)
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e-3 / steps



def aux_rows(args):
  """`python bench.py --aux`: the two rows built around the forward pass
  measured to the same bar — inputs resident in HBM, HIP events on the launch stream, algorithmic
  bytes against the HBM peak, the CPU restatement (oracle, NumPy) timed beside it.  One JSON line
  per row; not part of the driver's contract line.
    projection        [120k points, 4 floats] -> range image [64, 2048, 5]; 16 B/point + 24 B/pixel
    confusion_matrix  labels + predictions of 32 scans 64x2048, NC 20 -> int64 [20, 20]; 8 B/pixel"""
  args.steps, args.warmup = max(args.steps, 200), max(args.warmup, 20)
  args.cpu_seconds = min(args.cpu_seconds, 5.0) or 1.0
  import torch
  from pclsegmentation_amd import engine as E
  from oracle import np_oracle as O      # CPU baseline leg only
  if not torch.cuda.is_available():
    raise SystemExit("bench.py --aux needs an MI355X; there is no CPU fallback")
  dev = E.torch_device(0)
  stream = torch.cuda.current_stream(dev)
  rng = np.random.default_rng(1234)

  # ---- spherical projection: a synthetic 64-beam sweep, KITTI field of view (3, -25) degrees
  H, W, M = 64, 2048, 120000
  yaw = rng.uniform(-np.pi, np.pi, M)
  pitch = np.deg2rad(rng.uniform(-24.5, 2.5, M))
  r = rng.uniform(2.0, 80.0, M)
  pts = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch),
                  rng.uniform(0, 1, M)], -1).astype(np.float32)
  d_pts = torch.from_numpy(pts).to(dev)
  d_img = torch.empty((H, W, 5), dtype=torch.float32, device=dev)
  d_idx = torch.empty((H, W), dtype=torch.int32, device=dev)
  d_scr = torch.empty((H * W,), dtype=torch.int64, device=dev)
  t = _timed(lambda: E.op_project(d_pts, M, H, W, 3.0, -25.0, -1.0, d_img, d_idx, d_scr, stream.cuda_stream),
            stream, args.steps, args.warmup)
  alg = M * 16 + H * W * 24
  n = 0
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < args.cpu_seconds:
    O.range_projection(pts, H, W, 3.0, -25.0)
    n += 1
  cpu = n / (time.perf_counter() - t0)
  print(json.dumps({
    "metric": "LiDAR sweeps/sec spherical projection (120k points -> 64x2048x5)", "value": round(1.0 / t, 1),
    "unit": "sweeps/s", "us_per_sweep": round(t * 1e6, 2), "dtype": "f32 (f64 angles)", "data": "synthetic",
    "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes": alg,
                 "note": "3 launches (init, atomicMin scatter, gather); 4.5 MB per sweep: launch-latency bound, not bandwidth"},
    "cpu_baseline": {"value": round(cpu, 2), "unit": "sweeps/s", "cores": 1, "kind": "port",
                     "sample": "%d sweeps, NumPy restatement of LaserScan.do_range_projection (oracle/np_oracle.py)" % n}}),
    flush=True)

  # ---- confusion matrix
  N, NC = 32, 20
  P = N * H * W
  labels = rng.integers(0, NC, P, dtype=np.int32)
  preds = np.where(rng.random(P) < 0.8, labels, rng.integers(0, NC, P, dtype=np.int32)).astype(np.int32)
  d_l, d_p = torch.from_numpy(labels).to(dev), torch.from_numpy(preds).to(dev)
  d_cm = torch.zeros((NC, NC), dtype=torch.int64, device=dev)
  t = _timed(lambda: E.op_confusion_matrix(d_l, d_p, P, NC, d_cm, stream.cuda_stream), stream, args.steps, args.warmup)
  alg = P * 8
  n = 0
  t0 = time.perf_counter()
  sample = 4 * H * W
  while time.perf_counter() - t0 < args.cpu_seconds:
    O.confusion_matrix(labels[:sample], preds[:sample], NC)
    n += 1
  cpu = n * 4 / (time.perf_counter() - t0)
  print(json.dumps({
    "metric": "LiDAR scans/sec confusion-matrix accumulation (64x2048, NC 20)", "value": round(N / t, 1),
    "unit": "scans/s", "us_per_batch": round(t * 1e6, 2), "dtype": "int32 -> int64 counts", "data": "synthetic",
    "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes": alg},
    "cpu_baseline": {"value": round(cpu, 1), "unit": "scans/s", "cores": 1, "kind": "port",
                     "sample": "%d x 4 scans, NumPy bincount restatement (oracle/np_oracle.py)" % n}}), flush=True)



def csrc_sha():
  """sha256 over the kernel / engine sources: ties a committed PMC traffic figure to the code it
  was measured on (bench.py refuses to print a stale one)."""
  import hashlib
  h = hashlib.sha256()
  for rel in ("pclsegmentation_amd/csrc/pclseg_kernels.h", "pclsegmentation_amd/csrc/pclseg_graph.h",
              "pclsegmentation_amd/csrc/pclseg_api.hip", "include/pclseg.h"):
    h.update(open(os.path.join(ROOT, rel), "rb").read())
  return h.hexdigest()[:16]


def visible_gpu_count():
  """GPUs this process tree may use, WITHOUT touching HIP (the launcher must stay free of any GPU state:
  its children are the only processes that initialise the device).  KFD topology nodes with SIMDs are GPUs;
  ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow the set."""
  import glob
  n = 0
  for node in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
    try:
      props = dict(l.split()[:2] for l in open(node).read().splitlines() if len(l.split()) >= 2)
      n += int(props.get("simd_count", "0")) > 0
    except OSError:
      pass
  for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
    v = os.environ.get(var)
    if v is not None:
      n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
  return n


def self_launch(args):
  """`python bench.py --gpus N` (N > 1) outside torchrun: start N fresh rank processes with
  torch.distributed.run and relay their exit status.  THIS process never touches the GPU — the device
  count comes from sysfs and the environment, not from HIP — and the children are new processes, not an
  exec of this one.  Each rank binds to its device and NUMA-local cores before its first GPU call
  (pclsegmentation_amd.distributed.bind_rank)."""
  import socket
  import subprocess
  with socket.socket() as so:
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  if visible_gpu_count() < args.gpus:
    # fewer GPUs than ranks (e.g. a 1-GPU box): ranks share devices, and RCCL cannot put two
    # ranks on one device -> rendezvous / broadcast over gloo.  A functional check, not a scaling run.
    env["PCLSEG_DIST_BACKEND"] = "gloo"
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def build_engine(P, D, synthetic_weights, workload, dev_index, dev, rank, micro_batch=0, flags=0, batch=0):
  """Rank 0 creates the (seeded) weights, folds and packs them once; the packed device blob is
  broadcast (RCCL over xGMI) and the other ranks import it — pclsegmentation_amd.distributed.broadcast_engine."""
  model_name, config_name, h, w, wl_batch, pvalid, bound = WORKLOADS[workload]
  mc, model = P.load_model_config(model_name, config_name, height=h, width=w, device=dev_index,
                                  micro_batch=micro_batch)
  weights = None
  if rank == 0:
    weights = synthetic_weights(model.weight_spec(), 4321)
    model.set_weights(weights)
  eng = D.broadcast_engine(model, h, w, flags, src=0, device=dev)
  return mc, model, eng, weights, (batch or wl_batch)


def time_steps(torch, eng, E, scans, preds, batch, stream, steps, warmup, fence, repeats=1):
  """-> [(wall seconds, device milliseconds)] x repeats, each of EXACTLY `steps` forward_raw calls fenced
  (barrier + device synchronise) on both sides; `warmup` untimed steps before the first region."""
  def step():
    eng.forward_raw(scans, batch, preds, None, None, None, mem=E.MEM_DEVICE)
  for _ in range(warmup):
    step()
  out = []
  for _ in range(repeats):
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(steps):
      step()
    ev1.record(stream)
    fence()
    out.append((time.perf_counter() - t0, ev0.elapsed_time(ev1)))      # HIP events on the engine's stream
  return out


def roofline_of(bound, scans_per_s_dev, info):
  alg_bytes, alg_flops = info["alg_bytes_per_scan"], 2 * info["alg_macs_per_scan"]
  hbm_gbs = scans_per_s_dev * alg_bytes / 1e9
  mfma_tf = scans_per_s_dev * alg_flops / 1e12
  if bound == "hbm":
    roof = {"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(hbm_gbs / HBM_PEAK_GBS, 4), "traffic": None}
  else:
    # useful (algorithmic) FLOP/s against the dense f16 MFMA peak divided by the 3 products the
    # split-f16 arithmetic spends per multiply-accumulate
    peak = F16_MFMA_PEAK_TF / 3.0
    roof = {"bound": "mfma", "achieved": round(mfma_tf, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(mfma_tf / peak, 4), "traffic": None}
  roof["kernel"] = "all kernels of one forward step (HIP events on the engine stream)"
  roof["scans_per_s_device_clock"] = round(scans_per_s_dev, 2)
  roof["alg_bytes_per_scan"] = alg_bytes
  roof["alg_flops_per_scan"] = alg_flops
  roof["other"] = {"hbm_GBs": round(hbm_gbs, 1), "alg_TFLOPs": round(mfma_tf, 2),
                   "frac_of_f16_mfma_peak_div3": round(mfma_tf / (F16_MFMA_PEAK_TF / 3.0), 4),
                   "frac_of_f32_mfma_peak": round(mfma_tf / F32_MFMA_PEAK_TF, 4)}
  return roof


def _latest_traffic_json():
  """profiles/rNN_traffic.json of the highest round (attach_traffic quotes it only for the binary it was measured on)."""
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
  return files[-1] if files else os.path.join(ROOT, "profiles", "r04_traffic.json")


TRAFFIC_JSON = os.environ.get("PCLSEG_TRAFFIC_JSON") or _latest_traffic_json()
REPEATS = 5       # the timed region (K steps, fenced) is measured this many times: `value` = median, `spread` = [min, max]


def attach_traffic(roof, workload, scans_per_s_dev, info):
  """Measured HBM-side bytes per scan (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
  same command, committed under profiles/) — quoted only when the committed figure was measured on
  exactly these kernel sources — and the PHYSICAL roofline fractions that follow from it: the ALG
  fraction counts every reference module's input and output, which fusion no longer moves."""
  if not os.path.exists(TRAFFIC_JSON):
    return
  t = json.load(open(TRAFFIC_JSON))
  w = t.get("workloads", {}).get(workload)
  if not w:
    return
  from pclsegmentation_amd import engine as E
  name = os.path.relpath(TRAFFIC_JSON, ROOT)
  if E.build_sha() != csrc_sha():
    roof["traffic_note"] = ("the loaded libpclseg.so was built from sources with sha %s, the sources beside it hash to %s: "
                            "no profile figure is attributed to this binary" % (E.build_sha(), csrc_sha()))
    return
  if t.get("csrc_sha") != csrc_sha():
    roof["traffic_note"] = ("%s was measured on csrc sha %s, this build is %s: stale "
                            "figure withheld" % (name, t.get("csrc_sha"), csrc_sha()))
    return
  roof["traffic"] = int(w["hbm_bytes_per_scan"])
  roof["traffic_unit"] = ("HBM-side bytes per scan, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE), "
                          "%s, csrc sha %s = pclseg_build_sha() of the binary that ran" % (name, t["csrc_sha"]))
  if t.get("carried_over"):
    roof["traffic_unit"] += "; measured on csrc sha %s — %s" % (t.get("measured_on_csrc_sha"), t["carried_over"])
  roof["physical"] = {
    "hbm_frac": round(w["hbm_bytes_per_scan"] * scans_per_s_dev / (HBM_PEAK_GBS * 1e9), 4),
    "mfma_frac": round(3 * 2 * info["alg_macs_per_scan"] * scans_per_s_dev / (F16_MFMA_PEAK_TF * 1e12), 4),
    "note": "hbm_frac = measured traffic x scans/s / 8 TB/s; mfma_frac = 3 f16 products per MAC x ALG_FLOPS x scans/s / 2.5 PFLOP/s"}


def parity_check(P, E, synthetic_weights, synthetic_scans, workload, dev, dev_index):
  """One scan of `workload` at its full size against the float64 oracle (oracle/np_oracle.py), both
  arithmetic modes, OUTSIDE any timed region: the evidence that the numbers in this line are numbers of
  a correct forward pass (north_star: class IDs identical where decided, logits within 1e-3)."""
  import torch
  from oracle import np_oracle as O      # the checker, never the thing measured
  model_name, config_name, h, w, _, pvalid, _ = WORKLOADS[workload]
  mc, model = P.load_model_config(model_name, config_name, height=h, width=w, device=dev_index)
  model.set_weights(synthetic_weights(model.weight_spec(), 4321))
  raw = synthetic_scans(1, h, w, mc.INPUT_MEAN, mc.INPUT_STD, pvalid, seed=1234)
  none_index = mc.CLASSES.index("None")
  lidar, omask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
  kw = {"num_layers": mc.get("NUM_LAYERS")} if model_name != "squeezesegv2" else {}
  t0 = time.perf_counter()
  _, opred, ologits = O.forward(model.arch_name(), model.weights, lidar, omask, none_index, dtype=np.float64, **kw)
  oracle_s = time.perf_counter() - t0
  srt = np.sort(ologits[0], -1)
  decided = (srt[..., -1] - srt[..., -2]) > 2e-3
  out = {"workload": workload, "scan": "synthetic seed 1234, scan 0, %dx%d" % (h, w),
         "oracle": "oracle/np_oracle.py float64 (%.1f s)" % oracle_s,
         "decided_pixels": int(decided.sum()), "pixels": int(decided.size),
         "max_abs_logit": round(float(np.abs(ologits).max()), 3)}
  d_raw = torch.from_numpy(raw).to(dev)
  for key, flags in (("f16x3", 0), ("f32", E.FLAG_EXACT_F32)):
    eng = model.engine(h, w, flags)
    preds = torch.empty((1, h, w), dtype=torch.int32, device=dev)
    logits = torch.empty((1, h, w, mc.NUM_CLASS), dtype=torch.float32, device=dev)
    eng.set_stream(E.stream_handle(dev))
    eng.forward_raw(d_raw, 1, preds, None, logits, None, mem=E.MEM_DEVICE)
    eng.sync()
    pr, lg = preds.cpu().numpy()[0], logits.cpu().numpy()[0].astype(np.float64)
    out["max_abs_logit_err_" + key] = float("%.3g" % np.abs(lg - ologits[0]).max())
    out["decided_identical_" + key] = bool(np.array_equal(pr[decided], opred[0][decided]))
    out["masked_are_none_" + key] = bool((pr[~omask[0]] == none_index).all())
    model._drop_engines()
  out["decided_identical"] = out["decided_identical_f16x3"] and out["decided_identical_f32"]
  return out


def c1_gpu_leg(P, E, synthetic_weights, dev_index):
  """BASELINE.json configs[0] (C1) on the GPU, in the reference's own loop shape (inference.py:44-75): the 32
  real 32x240 scans of dataset_samples/sample_dataset/train (committed fixture), ONE scan per call, host
  buffers in and out, every call synchronous (PCLSEG_MEM_HOST) — what a drop-in user of inference.py sees per
  file, next to cpu_baseline.legs.c1_real_32x240_batch1_*.  At this size a call is `launches_per_scan` kernel
  launches of a few microseconds each: the figure is launch/round-trip latency, not throughput."""
  real = os.path.join(ROOT, "tests", "golden", "c1_sample_dataset_train_32x240.npz")
  if not os.path.exists(real):
    return None
  import torch
  mc1, m1 = P.load_model_config("squeezesegv2", "squeezesegv2", device=dev_index)
  m1.set_weights(synthetic_weights(m1.weight_spec(), 4321))
  raw = np.ascontiguousarray(np.load(real)["raw"], dtype=np.float32)     # [32, 32, 240, 5]
  n, h, w, _ = raw.shape
  eng = m1.engine(h, w)
  info = E.plan(eng.desc)

  def loop(scans, preds, passes):
    times = []
    for _ in range(passes):
      t0 = time.perf_counter()
      for i in range(n):
        eng.forward_raw(scans[i:i + 1], 1, preds[i:i + 1], None, None, None, mem=E.MEM_HOST)
      times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2]

  preds = np.empty((n, h, w), np.int32)
  loop(raw, preds, 2)
  t_page = loop(raw, preds, 7)
  p_raw = torch.from_numpy(raw).pin_memory()
  p_preds = torch.empty((n, h, w), dtype=torch.int32).pin_memory()
  loop(p_raw, p_preds, 2)
  t_pin = loop(p_raw, p_preds, 7)
  same = bool(np.array_equal(p_preds.numpy(), preds))
  b_preds = np.empty((n, h, w), np.int32)
  eng.forward_raw(raw, n, b_preds, None, None, None, mem=E.MEM_HOST)
  t0 = time.perf_counter()
  for _ in range(5):
    eng.forward_raw(raw, n, b_preds, None, None, None, mem=E.MEM_HOST)
  t_batch = (time.perf_counter() - t0) / 5
  out = {"value": round(n / t_page, 1), "unit": "scans/s", "us_per_scan": round(1e6 * t_page / n, 1),
         "launches_per_scan": info["num_ops"] + 1,
         "sample": "the 32 real 32x240 scans of the reference's sample_dataset/train, batch 1, synchronous PCLSEG_MEM_HOST calls "
                   "on pageable NumPy buffers (median of 7 passes): the reference's loop, inference.py:44-75",
         "page_locked": {"value": round(n / t_pin, 1), "us_per_scan": round(1e6 * t_pin / n, 1)},
         "one_call_batch32": {"value": round(n / t_batch, 1), "us_per_scan": round(1e6 * t_batch / n, 1),
                              "identical_to_batch1": bool(np.array_equal(b_preds, preds))},
         "page_locked_identical": same}
  m1._drop_engines()
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=100)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--workload", default="ssv2_64x2048", choices=sorted(WORKLOADS))
  ap.add_argument("--batch", type=int, default=0, help="scans per GPU per step (0 = workload default)")
  ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                  help="weak: every GPU runs --batch scans per step (default); strong: --global-batch scans per "
                       "step are sharded over the GPUs by contiguous ranges (BASELINE configs[3]: 256 over 8)")
  ap.add_argument("--global-batch", type=int, default=256, help="scans per step over all GPUs with --scaling strong")
  ap.add_argument("--micro-batch", type=int, default=0)
  ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline budget; 0 disables")
  ap.add_argument("--no-secondary", action="store_true",
                  help="skip the secondary rows (Darknet workloads, exact-f32, host boundary, parity check)")
  ap.add_argument("--aux", action="store_true", help="measure the projection / confusion-matrix rows instead")
  args = ap.parse_args()
  if args.aux:
    return aux_rows(args)
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    raise SystemExit(self_launch(args))

  # (nothing above has touched the GPU; init_process_group pins the rank to its NUMA-local cores first)
  import pclsegmentation_amd as P
  from pclsegmentation_amd import distributed as D
  rank, local_rank, world = D.env_world()
  if world > 1:
    D.bind_rank(local_rank)
  import torch
  from pclsegmentation_amd import engine as E
  from pclsegmentation_amd.nets.weights import synthetic_weights
  from pclsegmentation_amd.utils.synthetic import synthetic_scans

  rank, local_rank, world = D.init_process_group()
  if world != args.gpus:
    raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
  dev_index = local_rank % torch.cuda.device_count()   # identity on a node with one GPU per rank
  torch.cuda.set_device(dev_index)
  dev = E.torch_device(dev_index)
  stream = torch.cuda.current_stream(dev)

  coll = D.collectives_active()   # > 1 rank, or a forced one-rank group (PCLSEG_FORCE_COLLECTIVES=1: test aid)

  def fence():
    torch.cuda.synchronize(dev)
    if coll:
      torch.distributed.barrier()
    torch.cuda.synchronize(dev)

  def run_workload(workload, steps, warmup, flags=0, batch=0, repeats=1):
    """Build the engine of `workload`, time `steps` steps (`repeats` fenced regions of `steps` steps each, every
    region reduced with MAX over the ranks; the MEDIAN region is reported); -> dict."""
    model_name, config_name, h, w, _, pvalid, bound = WORKLOADS[workload]
    mc, model, eng, weights, batch = build_engine(P, D, synthetic_weights, workload, dev_index, dev, rank,
                                                  args.micro_batch, flags, batch)
    global_batch = batch * world
    if args.scaling == "strong":     # fixed total work: this rank's contiguous share of the global batch
      global_batch = args.global_batch
      lo, hi = D.shard_range(global_batch, rank, world)
      batch = hi - lo
      if batch <= 0:
        raise SystemExit("--global-batch %d leaves rank %d without scans" % (global_batch, rank))
    eng.set_stream(stream.cuda_stream)
    info = E.plan(eng.desc)
    scans = torch.from_numpy(synthetic_scans(batch, h, w, mc.INPUT_MEAN, mc.INPUT_STD, pvalid,
                                             seed=1234 + rank)).to(dev)
    preds = torch.empty((batch, h, w), dtype=torch.int32, device=dev)
    regions = time_steps(torch, eng, E, scans, preds, batch, stream, steps, warmup, fence, repeats)
    eng.sync()   # also reports a split-f16 range overflow (PCLSEG_ERR_RANGE) instead of timing garbage
    if coll:
      red_dev = dev if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
      t = torch.tensor(regions, dtype=torch.float64, device=red_dev)
      torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)   # MAX over ranks, per region
      regions = [(float(a), float(b)) for a, b in t.cpu().tolist()]
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, dev_ms = regions[order[len(order) // 2]]                      # the median region
    res = {"mc": mc, "model": model, "eng": eng, "weights": weights, "info": info, "batch": batch,
           "global_batch": global_batch,
           "scans": scans, "preds": preds, "elapsed": elapsed, "dev_ms": dev_ms, "h": h, "w": w,
           "pvalid": pvalid, "bound": bound, "model_name": model_name,
           "scans_per_s": global_batch * steps / elapsed,
           "dev_scans_per_s": batch * steps / (dev_ms * 1e-3),
           "region_scans_per_s": [round(global_batch * steps / r[0], 2) for r in regions]}
    return res

  r = run_workload(args.workload, args.steps, args.warmup, batch=args.batch, repeats=REPEATS)
  if rank == 0:
    roof = roofline_of(r["bound"], r["dev_scans_per_s"], r["info"])
    attach_traffic(roof, args.workload, r["dev_scans_per_s"], r["info"])
    mc = r["mc"]
    out = {
      "metric": "LiDAR scans/sec (64x2048) SqueezeSegV2 inference" if args.workload == "ssv2_64x2048"
                else "LiDAR scans/sec %s inference" % args.workload,
      "value": round(r["scans_per_s"], 2), "unit": "scans/s", "n_gpus": world, "steps": args.steps,
      "warmup": args.warmup, "ms_per_step": round(1e3 * r["elapsed"] / args.steps, 3),
      "spread": [min(r["region_scans_per_s"]), max(r["region_scans_per_s"])],
      "repeats": {"regions": REPEATS, "steps_each": args.steps, "scans_per_s": r["region_scans_per_s"],
                  "note": "the timed region of exactly --steps steps (barrier + synchronise on both sides, MAX over ranks) "
                          "is measured %d times back to back after ONE warm-up; value / ms_per_step = the median region" % REPEATS},
      "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
      "dtype": "f32 (f16x3 products)", "data": "synthetic",
      "config": {"workload": args.workload, "model": r["model_name"], "shape": [r["h"], r["w"]],
                 "num_class": mc.NUM_CLASS, "batch_per_gpu": r["batch"], "global_batch": r["global_batch"],
                 "math": "storage, accumulation and outputs float32; products on split-f16 operands (hi*hi + hi*lo + "
                         "lo*hi on v_mfma_f32_16x16x32_f16): weights pre-scaled per 16-channel tile so every weight "
                         "keeps 22 bits, activations 22 bits where |v| >= 1/8 and an absolute 2^-25 below; measured "
                         "logit error in parity_check",
                 "micro_batch": r["info"]["micro_batch"], "lanes": int(os.environ.get("PCLSEG_LANES", "3")),
                 "parallelism": "batch-sharded x%d (%s scaling, weights folded+packed once and broadcast)" % (world, args.scaling)},
      "roofline": roof,
      "build": {"binary_sha": E.build_sha(), "sources_sha": csrc_sha(), "binary_matches_sources": E.build_sha() == csrc_sha()},
    }
    if coll:
      out["config"]["collectives"] = {"backend": torch.distributed.get_backend(), "world": world,
                                      "timing_reduced_on": "device" if torch.distributed.get_backend() == "nccl" else "host"}
    if os.environ.get("PCLSEG_DIST_BACKEND") == "gloo" and world > 1:
      out["config"]["note"] = "ranks share GPUs (fewer devices than ranks): functional check, not a scaling run"
  if world == 1 and not args.no_secondary:
    batch, h, w = r["batch"], r["h"], r["w"]
    # ---- the same step when the boundary hands over HOST buffers (never `value`): raw scans cross
    # PCIe in, int32 predictions cross it back, page-locked buffers, per-micro-batch copies on the
    # lane streams overlapped with the kernels of the other lanes
    eng = r["eng"]
    h_scans = torch.empty((batch, h, w, 5), dtype=torch.float32).pin_memory()
    h_scans.copy_(r["scans"].cpu())
    h_preds = torch.empty((batch, h, w), dtype=torch.int32).pin_memory()
    hs = min(args.steps, 30)
    for _ in range(3):
      eng.forward_raw(h_scans, batch, h_preds, None, None, None, mem=E.MEM_HOST_ASYNC)
    eng.sync()
    t1 = time.perf_counter()
    for _ in range(hs):      # enqueue-only calls, one wait at the end: the same call pattern as the device-resident loop
      eng.forward_raw(h_scans, batch, h_preds, None, None, None, mem=E.MEM_HOST_ASYNC)
    eng.sync()
    pinned = batch * hs / (time.perf_counter() - t1)
    same = bool(torch.equal(h_preds, r["preds"].cpu()))
    for _ in range(2):
      eng.forward_raw(h_scans, batch, h_preds, None, None, None, mem=E.MEM_HOST)
    t1 = time.perf_counter()
    for _ in range(hs):      # synchronous calls: the GPU idles while the CPU enqueues the next batch
      eng.forward_raw(h_scans, batch, h_preds, None, None, None, mem=E.MEM_HOST)
    pinned_sync = batch * hs / (time.perf_counter() - t1)
    p_scans, p_preds = h_scans.numpy().copy(), np.empty((batch, h, w), np.int32)   # pageable
    for _ in range(2):
      eng.forward_raw(p_scans, batch, p_preds, None, None, None, mem=E.MEM_HOST)
    t1 = time.perf_counter()
    for _ in range(hs):
      eng.forward_raw(p_scans, batch, p_preds, None, None, None, mem=E.MEM_HOST)
    pageable = batch * hs / (time.perf_counter() - t1)
    out["host_boundary"] = {
      "value": round(pinned_sync, 1), "unit": "scans/s", "frac_of_device_resident": round(pinned_sync / r["scans_per_s"], 3),
      "note": "PCLSEG_MEM_HOST, page-locked host buffers in and out over PCIe, every call waits for its outputs; "
              "per micro-batch a DMA upload that runs ahead of the lane or a copy kernel on the lane, predictions "
              "written by the head straight into the caller's buffer, overlapped with the other lanes' compute; "
              "predictions identical to the device-resident run: %s" % same,
      "enqueue_only_calls": {"value": round(pinned, 1),
                             "note": "PCLSEG_MEM_HOST_ASYNC (calls enqueue, one pclseg_sync at the end)"},
      "pageable": {"value": round(pageable, 1), "note": "PCLSEG_MEM_HOST, pageable NumPy buffers through the library's pinned bounce slabs"}}
    # ---- the headline workload with exact float32 products (PCLSEG_FLAG_EXACT_F32)
    del eng
    r["model"]._drop_engines()
    x = run_workload(args.workload, 5, 2, flags=E.FLAG_EXACT_F32, batch=args.batch)
    out["exact_f32"] = {"value": round(x["scans_per_s"], 1), "unit": "scans/s",
                        "ms_per_step": round(1e3 * x["elapsed"] / 5, 3),
                        "note": "same workload, every product on v_mfma_f32_16x16x4_f32 (bit-exact float32)"}
    x["model"]._drop_engines()
    del x
    # ---- the other single-GPU configurations of BASELINE.json (parity-tested in tests/, timed here;
    # their bound is the matrix cores)
    out["secondary"] = []
    for wl, st, wu in SECONDARY:
      if wl == args.workload:
        continue
      y = run_workload(wl, st, wu)
      roof_y = roofline_of(y["bound"], y["dev_scans_per_s"], y["info"])
      attach_traffic(roof_y, wl, y["dev_scans_per_s"], y["info"])
      out["secondary"].append({
        "workload": wl, "value": round(y["scans_per_s"], 1), "unit": "scans/s", "steps": st, "warmup": wu,
        "batch": y["batch"], "ms_per_step": round(1e3 * y["elapsed"] / st, 3), "roofline": roof_y})
      y["model"]._drop_engines()
      del y
    # ---- BASELINE configs[0] in the reference's loop shape, on the GPU
    try:      # (a leg added this round: whatever goes wrong in it must not cost the driver its bench line)
      c1 = c1_gpu_leg(P, E, synthetic_weights, dev_index)
    except Exception as e:   # noqa: BLE001
      c1 = {"error": "%s: %s" % (type(e).__name__, e)}
    if c1:
      out["c1_gpu"] = c1
    # ---- parity evidence of this very build, outside every timed region
    out["parity_check"] = [parity_check(P, E, synthetic_weights, synthetic_scans, wl, dev, dev_index)
                           for wl in PARITY_WORKLOADS]
  if rank == 0:
    if world == 1 and args.cpu_seconds > 0:
      out["cpu_baseline"] = cpu_baseline(r["model_name"], r["mc"], r["weights"], r["h"], r["w"], r["pvalid"],
                                         args.cpu_seconds)
    print(json.dumps(out), flush=True)
  if coll:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
  main()
