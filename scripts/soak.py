#!/usr/bin/env python3
"""Determinism soak under full multi-lane load: the same batch N times through device-resident,
synchronous-host and enqueue-only-host calls; every result must be bit-identical to the first.
usage: soak.py [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pclsegmentation_amd as P  # noqa: E402
from pclsegmentation_amd import engine as E  # noqa: E402
from pclsegmentation_amd.nets.weights import synthetic_weights  # noqa: E402
from pclsegmentation_amd.utils.synthetic import synthetic_scans  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for name, h, w, batch in (("squeezesegv2", 64, 2048, 32), ("squeezesegv2", 30, 208, 13), ("darknet21", 32, 1024, 16)):
  mc, model = P.load_model_config(name, name, height=h, width=w, device=0)
  model.set_weights(synthetic_weights(model.weight_spec(), 4321))
  eng = model.engine(h, w, 0)
  dev = torch.device("cuda", 0)
  eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
  h_scans = torch.from_numpy(synthetic_scans(batch, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=99)).pin_memory()
  d_scans = h_scans.to(dev)
  ref = torch.empty((batch, h, w), dtype=torch.int32, device=dev)
  eng.forward_raw(d_scans, batch, ref, None, None, None, mem=E.MEM_DEVICE)
  eng.sync()
  ref = ref.cpu()
  d_out = [torch.empty((batch, h, w), dtype=torch.int32, device=dev) for _ in range(4)]
  h_out = [torch.empty((batch, h, w), dtype=torch.int32).pin_memory() for _ in range(4)]
  n_it = iters if h * w * batch > 1 << 20 else iters * 4
  for i in range(n_it):
    k = i % 4
    mode = (i // 4) % 3
    if mode == 0:
      eng.forward_raw(d_scans, batch, d_out[k], None, None, None, mem=E.MEM_DEVICE)
    elif mode == 1:
      eng.forward_raw(h_scans, batch, h_out[k], None, None, None, mem=E.MEM_HOST_ASYNC)
    else:
      eng.forward_raw(h_scans, batch, h_out[k], None, None, None, mem=E.MEM_HOST)
    if k == 3:
      eng.sync()
      outs = d_out if mode == 0 else h_out
      for o in outs:
        if not torch.equal(o.cpu(), ref):
          bad += 1
          print("MISMATCH %s %dx%d iteration %d mode %d: %d pixels" % (name, h, w, i, mode, int((o.cpu() != ref).sum())))
  model._drop_engines()
  print("%s %dx%d batch %d: %d iterations done" % (name, h, w, batch, n_it))
print("soak:", "FAILED, %d mismatching results" % bad if bad else "all results bit-identical")
sys.exit(1 if bad else 0)
