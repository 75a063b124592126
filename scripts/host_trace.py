#!/usr/bin/env python3
"""Workload for a copy + kernel timeline of the host-boundary modes (run under
rocprofv3 --kernel-trace --memory-copy-trace).   usage: host_trace.py sync|async|device [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pclsegmentation_amd as P  # noqa: E402
from pclsegmentation_amd import engine as E  # noqa: E402
from pclsegmentation_amd.nets.weights import synthetic_weights  # noqa: E402
from pclsegmentation_amd.utils.synthetic import synthetic_scans  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "async"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
batch, h, w = 32, 64, 2048
mc, model = P.load_model_config("squeezesegv2", "squeezesegv2", height=h, width=w, device=0)
model.set_weights(synthetic_weights(model.weight_spec(), 4321))
eng = model.engine(h, w, 0)
eng.set_stream(torch.cuda.current_stream(torch.device("cuda", 0)).cuda_stream)
h_scans = torch.from_numpy(synthetic_scans(batch, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=1234)).pin_memory()
h_preds = torch.empty((batch, h, w), dtype=torch.int32).pin_memory()
mem = {"async": E.MEM_HOST_ASYNC, "sync": E.MEM_HOST, "device": E.MEM_DEVICE}[mode]
src, dst = (h_scans.cuda(), torch.empty((batch, h, w), dtype=torch.int32, device="cuda")) if mode == "device" else (h_scans, h_preds)
for _ in range(steps):
  eng.forward_raw(src, batch, dst, None, None, None, mem=mem)
eng.sync()
