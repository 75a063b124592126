"""Rank process of tests/test_gpu_distributed.py (started by torch.distributed.run): the engine's
multi-GPU path as bench.py / a serving job uses it — one broadcast of the PACKED parameters, then every rank runs ITS
contiguous shard of the batch with no data-path collective and writes its predictions to a file.

usage: dist_worker.py <out_dir> <n_scans> <h> <w> [config]        (config: squeezesegv2 | squeezesegv2kitti)
Scan i of the job is pclsegmentation_amd.utils.synthetic.synthetic_scan_range(i, i+1, seed=99): a rank generates only
its own shard (256 scans of 64x2048 are 671 MB).  Logits are written for jobs of at most 64 scans; larger jobs
write the predictions of the whole shard and the logits of its first scan."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PVALID = {"squeezesegv2": 0.84, "squeezesegv2kitti": 0.78}


def main():
  out_dir, n, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
  config = sys.argv[5] if len(sys.argv) > 5 else "squeezesegv2"
  import torch
  import pclsegmentation_amd as P
  from pclsegmentation_amd import distributed as D
  from pclsegmentation_amd import engine as E
  from pclsegmentation_amd.nets.weights import synthetic_weights
  from pclsegmentation_amd.utils.synthetic import synthetic_scan_range
  rank, local_rank, world = D.init_process_group()
  dev_index = local_rank % torch.cuda.device_count()
  torch.cuda.set_device(dev_index)
  dev = torch.device("cuda", dev_index)
  mc, model = P.load_model_config("squeezesegv2", config, height=h, width=w, device=dev_index)
  if rank == 0:                                  # only rank 0 ever holds the Keras tensors:
    model.set_weights(synthetic_weights(model.weight_spec(), 4321))
  eng = D.broadcast_engine(model, h, w, src=0, device=dev)   # folded + packed once, one broadcast, import elsewhere
  assert rank == 0 or model.weights is None
  lo, hi = D.shard_range(n, rank, world)
  raw = synthetic_scan_range(lo, hi, h, w, mc.INPUT_MEAN, mc.INPUT_STD, PVALID[config], seed=99)  # this rank's shard
  scans = torch.from_numpy(raw).to(dev)
  preds = torch.empty((hi - lo, h, w), dtype=torch.int32, device=dev)
  n_logits = (hi - lo) if n <= 64 else min(1, hi - lo)
  logits = torch.empty((n_logits, h, w, mc.NUM_CLASS), dtype=torch.float32, device=dev)
  eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
  if hi > lo:
    if n_logits == hi - lo:
      eng.forward_raw(scans, hi - lo, preds, None, logits, None, mem=E.MEM_DEVICE)
    else:
      eng.forward_raw(scans, hi - lo, preds, None, None, None, mem=E.MEM_DEVICE)
      first = torch.empty((1, h, w), dtype=torch.int32, device=dev)
      eng.forward_raw(scans, 1, first, None, logits, None, mem=E.MEM_DEVICE)
  eng.sync()
  nccl = torch.distributed.get_backend() == "nccl"
  np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, hi=hi, preds=preds.cpu().numpy(),
           logits=logits.cpu().numpy(), backend=torch.distributed.get_backend(), world=world,
           engine_came_through_collective=bool(rank != 0 or (world == 1 and D.force_collectives())))
  if n <= 64:
    full = D.gather_predictions(preds if nccl else preds.cpu(), n, dst=0)
    if rank == 0:
      np.save(os.path.join(out_dir, "gathered.npy"), full.cpu().numpy())
  if nccl:     # a device-side reduction, as bench.py does with its timings
    t = torch.tensor([float(rank + 1), 2.0], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    assert float(t[0]) == world and float(t[1]) == 2.0
  torch.distributed.barrier()
  torch.distributed.destroy_process_group()


if __name__ == "__main__":
  main()
