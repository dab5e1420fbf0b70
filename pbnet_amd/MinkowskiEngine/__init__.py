"""`import pbnet_amd.MinkowskiEngine as ME` -- the subset of MinkowskiEngine that PBNet's hot path touches
(boundary B), rebuilt on hand-written gfx950 kernels.  Same names, same call shapes."""
from . import utils, modules  # noqa: F401
from .core import SparseTensor, CoordinateManager, cat  # noqa: F401
from .conv import (MinkowskiConvolution, MinkowskiConvolutionTranspose, MinkowskiLinear,  # noqa: F401
                   spconv_forward, pack_weight)
from .nn import (MinkowskiBatchNorm, MinkowskiReLU, MinkowskiPReLU, MinkowskiSigmoid, MinkowskiSoftmax,  # noqa: F401
                 MinkowskiGlobalAvgPooling, MinkowskiGlobalMaxPooling)
