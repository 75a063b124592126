from .SqueezeSegV2 import SqueezeSegV2
from .Darknet import Darknet
from .SegmentationNetwork import PCLSegmentationNetwork
