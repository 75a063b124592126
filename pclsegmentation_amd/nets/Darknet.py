"""Darknet-21 / Darknet-53 model object (reference: nets/Darknet.py:147-314).

BasicBlock / EncoderLayer / DecoderLayer (:29-138), the output-stride "stride play"
(:158-181, :215-231) and the shape-driven skip connections (:263-277) are resolved by the
engine's graph builder (csrc/pclseg_graph.h: build_darknet); NUM_LAYERS selects the block
counts ``model_blocks`` (:142-145).
"""
from .SegmentationNetwork import PCLSegmentationNetwork

model_blocks = {
  21: [1, 1, 2, 2, 1],
  53: [1, 2, 8, 8, 4],
}


class Darknet(PCLSegmentationNetwork):
  def __init__(self, mc, **kw):
    super(Darknet, self).__init__(mc, **kw)
    self.drop_rate = mc.get("DROP_RATE", 0.0)
    self.bn_momentum = mc.get("BN_MOMENTUM", 0.9)
    self.output_stride = mc.OUTPUT_STRIDE
    self.num_layers = mc.NUM_LAYERS
    if self.num_layers not in model_blocks:
      raise KeyError(self.num_layers)
    self.num_blocks = model_blocks[self.num_layers]

  def arch_name(self):
    return "darknet%d" % self.num_layers
