"""Rank process of tests/test_gpu_distributed.py (started by torch.distributed.run): the engine's
multi-GPU path as bench.py / a serving job uses it — one broadcast of the PACKED parameters, then every rank runs ITS
contiguous shard of the batch with no data-path collective and writes its predictions to a file.

usage: dist_worker.py <out_dir> <n_scans> <h> <w>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  out_dir, n, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
  import torch
  import pclsegmentation_amd as P
  from pclsegmentation_amd import distributed as D
  from pclsegmentation_amd import engine as E
  from pclsegmentation_amd.nets.weights import synthetic_weights
  from pclsegmentation_amd.utils.synthetic import synthetic_scans
  rank, local_rank, world = D.init_process_group()
  dev_index = local_rank % torch.cuda.device_count()
  torch.cuda.set_device(dev_index)
  dev = torch.device("cuda", dev_index)
  mc, model = P.load_model_config("squeezesegv2", "squeezesegv2", height=h, width=w, device=dev_index)
  if rank == 0:                                  # only rank 0 ever holds the Keras tensors:
    model.set_weights(synthetic_weights(model.weight_spec(), 4321))
  eng = D.broadcast_engine(model, h, w, src=0, device=dev)   # folded + packed once, one broadcast, import elsewhere
  assert rank == 0 or model.weights is None
  raw = synthetic_scans(n, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.84, seed=99)  # the job's whole batch
  lo, hi = D.shard_range(n, rank, world)
  scans = torch.from_numpy(raw[lo:hi]).to(dev)
  preds = torch.empty((hi - lo, h, w), dtype=torch.int32, device=dev)
  logits = torch.empty((hi - lo, h, w, mc.NUM_CLASS), dtype=torch.float32, device=dev)
  eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
  if hi > lo:
    eng.forward_raw(scans, hi - lo, preds, None, logits, None, mem=E.MEM_DEVICE)
  eng.sync()
  np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, hi=hi, preds=preds.cpu().numpy(),
           logits=logits.cpu().numpy())
  full = D.gather_predictions(preds.cpu() if torch.distributed.get_backend() == "gloo" else preds, n, dst=0)
  if rank == 0:
    np.save(os.path.join(out_dir, "gathered.npy"), full.cpu().numpy())
  torch.distributed.barrier()
  torch.distributed.destroy_process_group()


if __name__ == "__main__":
  main()
