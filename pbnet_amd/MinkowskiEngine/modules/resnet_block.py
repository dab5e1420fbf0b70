"""MinkowskiEngine.modules.resnet_block.BasicBlock (imported by the reference at network/Mink.py:11)."""
import torch.nn as nn

from ..conv import MinkowskiConvolution
from ..nn import MinkowskiBatchNorm, MinkowskiReLU, bn_act


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    FUSED_TRAIN = True          # native training path: the whole block as one autograd node (fused_train.basic_block)

    def forward(self, x):
        if self.FUSED_TRAIN and self.training:
            from ..fused_train import basic_block
            y = basic_block(self, x)
            if y is not None:
                return y
        # the reference's block (conv -> bn -> relu -> conv -> bn -> += residual -> relu); bn_act runs each bn with its
        # tail as one pass when the native training path is on, and the separate modules otherwise
        residual = x
        out = bn_act(self.norm1, self.conv1(x))
        out = self.conv2(out)
        if self.downsample is not None:
            residual = self.downsample(x)
        return bn_act(self.norm2, out, residual=residual)


class Bottleneck(nn.Module):
    """Present in ME and imported by Mink.py:11, but no reachable network uses it."""
    expansion = 4

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("Bottleneck networks are never instantiated by PBNet (SURVEY.md section 2 #1)")
