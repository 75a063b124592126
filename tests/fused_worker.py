"""Worker of tests/test_gpu_models.py::test_fused_squeeze_intermediates (run as a subprocess with
PCLSEG_FUSE_KEEP=1, a debug switch that keeps the expand -> next-squeeze fusion ON together with
PCLSEG_FLAG_KEEP_ACTIVATIONS): every fused squeeze output (pool -> squeeze, CAM -> squeeze and expand -> squeeze) and
the logits against the float64 oracle.
Prints one line per tensor and exits non-zero on a mismatch.

usage: fused_worker.py <H> <W>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pclsegmentation_amd as P  # noqa: E402
from pclsegmentation_amd import engine as E  # noqa: E402
from pclsegmentation_amd.utils.synthetic import synthetic_scans  # noqa: E402
from oracle import np_oracle as O  # noqa: E402

h, w = int(sys.argv[1]), int(sys.argv[2])
mc, model = P.load_model_config("squeezesegv2", "squeezesegv2", height=h, width=w)
model.init_weights(4321)
raw = synthetic_scans(2, h, w, mc.INPUT_MEAN, mc.INPUT_STD, 0.8, seed=7)
eng = model.engine(h, w, E.FLAG_KEEP_ACTIVATIONS)
preds = np.empty((2, h, w), np.int32)
logits = np.empty((2, h, w, mc.NUM_CLASS), np.float32)
eng.forward_raw(np.ascontiguousarray(raw), 2, preds, None, logits, None, mem=E.MEM_HOST)
lidar, mask = O.normalize_and_mask(raw, mc.INPUT_MEAN, mc.INPUT_STD)
taps = {}
O.forward("squeezesegv2", model.weights, lidar, mask, mc.CLASSES.index("None"), dtype=np.float64, taps=taps)
names = [t[0] for t in eng.tensors()]
W = O._W(model.weights, np.float64)
bad = 0
for src, dst in (("pool1", "fire2"), ("pool3", "fire4"), ("pool5", "fire6"),   # pool -> squeeze (pool_squeeze_kernel)
                 ("cam2", "fire3"),                                              # CAM -> squeeze (cam_kernel SQ)
                 ("fire4", "fire5"), ("fire6", "fire7"), ("fire7", "fire8"), ("fire8", "fire9"), ("fire9", "fire10"),
                 ("fire10", "fire11"), ("fire11", "fire12"), ("fire12", "fire13")):
  if src in names and eng.tensors()[names.index(src)][1][0] > 0:
    pass   # (the pair output tensor still exists in the plan but is never written when fused)
  want = O.relu(W.bn(W.conv(taps[src], dst + "/squeeze"), dst + "/squeeze_bn"))
  got = eng.read_tensor(names.index(dst + "/squeeze"))
  err = float(np.abs(got - want).max())
  print("%s/squeeze %s max err %.3g" % (dst, got.shape, err))
  bad += err > 1e-4 * max(1.0, float(np.abs(want).max()))
err = float(np.abs(logits - taps["logits"]).max())
print("logits max err %.3g" % err)
sys.exit(1 if bad or err > 1e-3 else 0)
