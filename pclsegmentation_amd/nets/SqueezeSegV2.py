"""SqueezeSegV2 model object (reference: nets/SqueezeSegV2.py:217-334).

The CAM / FIRE / FIREUP layers and the encoder-decoder graph of the reference
(:30-213, :285-325) are built natively by the engine's graph builder
(csrc/pclseg_graph.h: build_squeezesegv2) and run as HIP kernels; this class only binds
the config and the Keras-path-keyed weights to that graph.
"""
from .SegmentationNetwork import PCLSegmentationNetwork


class SqueezeSegV2(PCLSegmentationNetwork):
  def __init__(self, mc, **kw):
    super(SqueezeSegV2, self).__init__(mc, **kw)
    self.drop_rate = mc.get("DROP_RATE", 0.0)        # dropout is identity at inference
    self.l2 = mc.get("L2_WEIGHT_DECAY", 0.0)          # training-only
    self.bn_momentum = mc.get("BN_MOMENTUM", 0.99)    # training-only

  def arch_name(self):
    return "squeezesegv2"
