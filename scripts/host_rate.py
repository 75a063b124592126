#!/usr/bin/env python3
"""Host-boundary throughput of the headline workload (page-locked buffers): device-resident calls,
synchronous MEM_HOST calls, enqueue-only MEM_HOST_ASYNC calls.   usage: host_rate.py [steps]   (env knobs apply)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pclsegmentation_amd as P  # noqa: E402
from pclsegmentation_amd import engine as E  # noqa: E402
from pclsegmentation_amd.nets.weights import synthetic_weights  # noqa: E402
from pclsegmentation_amd.utils.synthetic import synthetic_scans  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
batch, h, w = 32, 64, 2048
mc, model = P.load_model_config("squeezesegv2", "squeezesegv2", height=h, width=w, device=0)
model.set_weights(synthetic_weights(model.weight_spec(), 4321))
eng = model.engine(h, w, 0)
dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
scans = torch.from_numpy(synthetic_scans(batch, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=1234)).to(dev)
preds = torch.empty((batch, h, w), dtype=torch.int32, device=dev)
h_scans = torch.empty((batch, h, w, 5), dtype=torch.float32).pin_memory()
h_scans.copy_(scans.cpu())
h_preds = torch.empty((batch, h, w), dtype=torch.int32).pin_memory()
res = {}
p_scans, p_preds = h_scans.numpy().copy(), np.empty((batch, h, w), np.int32)   # pageable
for name, mem, src, dst in (("device", E.MEM_DEVICE, scans, preds), ("sync", E.MEM_HOST, h_scans, h_preds),
                            ("async", E.MEM_HOST_ASYNC, h_scans, h_preds), ("pageable", E.MEM_HOST, p_scans, p_preds)):
  for _ in range(5):
    eng.forward_raw(src, batch, dst, None, None, None, mem=mem)
  eng.sync()
  torch.cuda.synchronize(dev)
  t = time.perf_counter()
  for _ in range(steps):
    eng.forward_raw(src, batch, dst, None, None, None, mem=mem)
  t_enq = time.perf_counter() - t      # time the calls themselves took (enqueue only, for device / async)
  eng.sync()
  torch.cuda.synchronize(dev)
  res[name] = batch * steps / (time.perf_counter() - t)
  res[name + "_call_ms"] = 1e3 * t_enq / steps
  assert np.array_equal(dst if isinstance(dst, np.ndarray) else dst.cpu().numpy(), preds.cpu().numpy())
print(" ".join("%s=%.4g" % kv for kv in res.items()), "sync/device=%.3f async/device=%.3f pageable/device=%.3f" % (
  res["sync"] / res["device"], res["async"] / res["device"], res["pageable"] / res["device"]))
