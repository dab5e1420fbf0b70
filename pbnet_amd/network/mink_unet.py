"""MinkUNet family on the MI355X kernels: the backbone of /root/reference/network/Mink.py (MinkUNetBase
:202-354 and its variants :357-419; factory Mink_unet :502-526).  Module and parameter names match the reference so
a state dict moves across unchanged (conv0p1s1.kernel, bn0.bn.weight, block1.0.conv1.kernel,
block2.0.downsample.0.kernel, convtr4p16s2.kernel, final_sematic.bias, ...).

Two execution paths over the same parameters:
  * module path  -- one kernel launch per ME module, BatchNorm/ReLU as separate feature passes; used in training
    mode and as the unfused cross-check;
  * fused path   -- eval mode only: BatchNorm folded into a per-channel scale/shift that the convolution applies in
    its epilogue together with ReLU and the residual add; skip concatenations are written in place (the encoder
    block stores its result directly into the right-hand columns of the decoder's concat slab, the transposed
    convolution into the left-hand columns), so no activation is touched twice.
"""
import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from ..MinkowskiEngine.conv import spconv_forward, _pad_vec
from ..MinkowskiEngine.modules.resnet_block import BasicBlock
from ..prof import section

# arch -> (blocks per stage, planes)   (Mink.py:357-419; BasicBlock families only)
SPECS = {
    "MinkUNet14A": ((1,) * 8, (32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet14B": ((1,) * 8, (32, 64, 128, 256, 128, 128, 128, 128)),
    "MinkUNet14C": ((1,) * 8, (32, 64, 128, 256, 192, 192, 128, 128)),
    "MinkUNet14D": ((1,) * 8, (32, 64, 128, 256, 384, 384, 384, 384)),
    "MinkUNet18A": ((2,) * 8, (32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet18B": ((2,) * 8, (32, 64, 128, 256, 128, 128, 128, 128)),
    "MinkUNet18D": ((2,) * 8, (32, 64, 128, 256, 384, 384, 384, 384)),
    "MinkUNet34A": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 64, 64)),
    "MinkUNet34B": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 64, 32)),
    "MinkUNet34C": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 96, 96)),
}
INIT_DIM = 32          # Mink.py:208
_DOWN = ("conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2")
_DOWN_BN = ("bn1", "bn2", "bn3", "bn4")
_UP = ("convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2")
_UP_BN = ("bntr4", "bntr5", "bntr6", "bntr7")


class MinkUNet(nn.Module):
    FUSE_EVAL = True

    def __init__(self, in_channels, out_channels, D=3, arch="MinkUNet34C"):
        super().__init__()
        assert D == 3
        self.arch = arch
        self.LAYERS, self.PLANES = SPECS[arch]
        P, L = self.PLANES, self.LAYERS
        self.inplanes = INIT_DIM
        self.conv0p1s1 = ME.MinkowskiConvolution(in_channels, self.inplanes, kernel_size=5, dimension=D)
        self.bn0 = ME.MinkowskiBatchNorm(self.inplanes)
        # encoder: k2s2 down convolution + residual stage, four times (Mink.py:226-251)
        for i in range(4):
            setattr(self, _DOWN[i], ME.MinkowskiConvolution(self.inplanes, self.inplanes, kernel_size=2, stride=2,
                                                            dimension=D))
            setattr(self, _DOWN_BN[i], ME.MinkowskiBatchNorm(self.inplanes))
            setattr(self, "block%d" % (i + 1), self._make_layer(P[i], L[i], D))
        # decoder: transposed k2s2 + skip concat + residual stage (Mink.py:253-280)
        skips = (P[2], P[1], P[0], INIT_DIM)
        for i in range(4):
            setattr(self, _UP[i], ME.MinkowskiConvolutionTranspose(self.inplanes, P[4 + i], kernel_size=2, stride=2,
                                                                   dimension=D))
            setattr(self, _UP_BN[i], ME.MinkowskiBatchNorm(P[4 + i]))
            self.inplanes = P[4 + i] + skips[i]
            setattr(self, "block%d" % (i + 5), self._make_layer(P[4 + i], L[4 + i], D))
        self.final_sematic = ME.MinkowskiConvolution(P[7], out_channels, kernel_size=1, bias=True, dimension=D)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.weight_initialization()
        self._fold_cache = {}

    def _make_layer(self, planes, blocks, D):
        """Mink.py:75-107 with stride 1: a 1x1 conv + BN shortcut iff the channel count changes."""
        downsample = None
        if self.inplanes != planes:
            downsample = nn.Sequential(ME.MinkowskiConvolution(self.inplanes, planes, kernel_size=1, stride=1, dimension=D),
                                       ME.MinkowskiBatchNorm(planes))
        layers = [BasicBlock(self.inplanes, planes, stride=1, downsample=downsample, dimension=D)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes, stride=1, dimension=D) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def weight_initialization(self):
        """Mink.py:66-73: kaiming-normal(fan_out) on MinkowskiConvolution kernels (transposed ones keep ME's default
        uniform init: they are a different class), BN weight 1 / bias 0."""
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):
                ME.utils.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    # ---------------------------------------------------------------------------------------------------------
    def forward(self, x):
        if self.FUSE_EVAL and not self.training and not torch.is_grad_enabled():
            return self._forward_fused(x)
        return self._forward_modules(x)

    def _forward_modules(self, x):
        """Mink.py:291-354, module by module."""
        out = self.relu(self.bn0(self.conv0p1s1(x)))
        skips = [out]
        for i in range(4):
            out = self.relu(getattr(self, _DOWN_BN[i])(getattr(self, _DOWN[i])(out)))
            out = getattr(self, "block%d" % (i + 1))(out)
            if i < 3:
                skips.append(out)
        for i in range(4):
            out = self.relu(getattr(self, _UP_BN[i])(getattr(self, _UP[i])(out)))
            out = ME.cat(out, skips[3 - i])
            out = getattr(self, "block%d" % (i + 5))(out)
        return self.final_sematic(out)

    # ---------------------------------------------------------------------------------------------------------
    def _fold(self, bn, cout_p):
        """Eval-mode BatchNorm as y = x*scale + shift (fp32, padded to the kernel's channel tile)."""
        b = bn.bn
        key = (id(bn), b.weight._version, b.bias._version, b.running_mean._version, b.running_var._version, cout_p)
        hit = self._fold_cache.get(id(bn))
        if hit is None or hit[0] != key:
            scale = b.weight.detach().float() / torch.sqrt(b.running_var.float() + b.eps)
            shift = b.bias.detach().float() - b.running_mean.float() * scale
            hit = (key, _pad_vec(scale, cout_p, 1.0), _pad_vec(shift, cout_p, 0.0))
            self._fold_cache[id(bn)] = hit
        return hit[1], hit[2]

    def _cbr(self, conv, bn, feats, nbr, n_out, relu=True, residual=None, out=None):
        packed = conv._cache.get(conv.kernel, feats.dtype)
        scale, shift = self._fold(bn, packed[3])
        return spconv_forward(feats, nbr, n_out, packed, scale=scale, shift=shift, residual=residual, relu=relu, out=out)

    def _stage_fused(self, stage, feats, nbr, n, out=None):
        nb = len(stage)
        for bi, blk in enumerate(stage):
            h = self._cbr(blk.conv1, blk.norm1, feats, nbr, n)
            res = feats
            if blk.downsample is not None:
                res = self._cbr(blk.downsample[0], blk.downsample[1], feats, None, n, relu=False)
            feats = self._cbr(blk.conv2, blk.norm2, h, nbr, n, relu=True, residual=res, out=out if bi == nb - 1 else None)
        return feats

    def _forward_fused(self, x):
        cm = x.coordinate_manager
        assert x.tensor_stride == 1
        P = self.PLANES
        dt, dev = x.F.dtype, x.F.device
        with section("unet.maps"):
            n = {s: cm.num_rows(s) for s in (1, 2, 4, 8, 16)}
            k3 = {s: cm.kernel_map(s, 3) for s in (1, 2, 4, 8, 16)}
            k5 = cm.kernel_map(1, 5)
            ups = {s: cm.up_map(s) for s in (2, 4, 8, 16)}
        skip_c = (INIT_DIM, P[0], P[1], P[2])                      # channels of out_p1, out_b1p2, out_b2p4, out_b3p8
        up_c = (P[7], P[6], P[5], P[4])                            # transposed-conv channels landing at stride 1,2,4,8
        strides = (1, 2, 4, 8)
        # concat slabs of the decoder: [up | skip] per stride (Mink.py:323,331,339,347)
        slab = {s: torch.empty(n[s], up_c[i] + skip_c[i], dtype=dt, device=dev) for i, s in enumerate(strides)}
        skip_view = {s: slab[s][:, up_c[i]:] for i, s in enumerate(strides)}
        up_view = {s: slab[s][:, :up_c[i]] for i, s in enumerate(strides)}

        with section("unet.stem"):
            cur = self._cbr(self.conv0p1s1, self.bn0, x.F, k5, n[1], out=skip_view[1])
        s = 1
        _sec = section("unet.encoder"); _sec.__enter__()
        for i in range(4):
            cur = self._cbr(getattr(self, _DOWN[i]), getattr(self, _DOWN_BN[i]), cur, cm.down_map(s), n[2 * s])
            s *= 2
            cur = self._stage_fused(getattr(self, "block%d" % (i + 1)), cur, k3[s], n[s],
                                    out=skip_view[s] if s < 16 else None)
        _sec.__exit__(None, None, None)
        _sec = section("unet.decoder"); _sec.__enter__()
        for i in range(4):
            self._cbr(getattr(self, _UP[i]), getattr(self, _UP_BN[i]), cur, ups[s], n[s // 2], out=up_view[s // 2])
            s //= 2
            cur = self._stage_fused(getattr(self, "block%d" % (i + 5)), slab[s], k3[s], n[s])
        _sec.__exit__(None, None, None)
        fs = self.final_sematic
        packed = fs._cache.get(fs.kernel, dt)
        out = spconv_forward(cur, None, n[1], packed, shift=_pad_vec(fs.bias, packed[3], 0.0))
        cout = fs.kernel.shape[-1]
        return ME.SparseTensor(out if out.shape[1] == cout else out[:, :cout], coordinate_manager=cm, tensor_stride=1)


def _variant(name):
    def init(self, in_channels, out_channels, D=3):
        MinkUNet.__init__(self, in_channels, out_channels, D, arch=name)
    return type(name, (MinkUNet,), {"__init__": init})


for _n in SPECS:
    globals()[_n] = _variant(_n)


def Mink_unet(in_channels=3, out_channels=20, D=3, arch="MinkUNet18A"):
    """Factory with the reference's signature (Mink.py:502-526)."""
    if arch not in SPECS:
        raise Exception("architecture not supported yet: {}".format(arch))
    return globals()[arch](in_channels, out_channels, D)
