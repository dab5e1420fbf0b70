"""Training-mode blocks of the MinkUNet as ONE autograd node each (/root/reference/network/Mink.py:293-350 and
MinkowskiEngine.modules.resnet_block.BasicBlock.forward): conv -> bn -> relu, and the residual block
conv -> bn -> relu -> conv -> bn -> (+ residual | + bn(1x1 conv)) -> relu.

Same kernels and the same arithmetic as the module-by-module path (pbn_spconv_forward for the convolutions and their
input gradients, pbn_bn_act_train_* for the normalisation with its tail, pbn_spconv_wgrad for the weight gradients); what
changes is the host side: a residual block is 5-7 native calls forward and 9-13 backward inside one Python function instead of
eight autograd nodes with their tensor wrappers -- the training step is host-bound, so this is where
its time goes.  The residual additions ride in the epilogues: `dx = dgrad(...) + d(residual)` is the convolution's residual
input, not a separate pass."""
import os

import torch

from .. import _native as N
from .conv import spconv_forward, wgrad_native
from .nn import _DT, _bn_workspace, _rows_ok


ENABLED = os.environ.get("PBN_TRAIN_FUSED", "1") == "1"      # False: every caller falls back to the module-by-module path


def usable(x, *norms):
    """The fused nodes serve the native training path only: CUDA slab with 16-byte rows, batch norms in training mode with a
    momentum and fp32 affine parameters, autograd on."""
    if not (ENABLED and torch.is_grad_enabled() and x.is_cuda and x.dtype in _DT and _rows_ok(x)):
        return False
    for nm in norms:
        bn = nm.bn
        if not (nm.NATIVE_TRAIN and nm.FUSE_ACT and bn.training and bn.momentum is not None and bn.weight is not None
                and bn.weight.dtype == torch.float32 and bn.num_features % 8 == 0):
            return False
    return True


def _bn_forward(x, norm, residual, relu):
    bn = norm.bn
    n, c = int(x.shape[0]), int(x.shape[1])
    y = torch.empty(n, c, dtype=x.dtype, device=x.device)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _bn_workspace(x.device, c)
    norm._tick()
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    N.check(N.lib().pbn_bn_act_train_forward(
        N.c_vp(x.data_ptr()), x.stride(0), n, c, _DT[x.dtype], N.ptr(bn.weight), N.ptr(bn.bias), float(bn.eps),
        float(bn.momentum), N.ptr(rm), N.ptr(rv), None if residual is None else N.c_vp(residual.data_ptr()),
        0 if residual is None else residual.stride(0), int(bool(relu)), N.c_vp(y.data_ptr()), c, N.ptr(mean), N.ptr(invstd),
        N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream()), "pbn_bn_act_train_forward")
    return y, mean, invstd


def _bn_backward(x, dy, y_mask, weight, mean, invstd, want_dres):
    """-> dx, dres (the masked dy; None when not wanted or when there is no mask: the caller uses dy itself), dweight, dbias"""
    n, c = int(x.shape[0]), int(x.shape[1])
    if dy.dtype != x.dtype or not _rows_ok(dy):
        dy = dy.to(x.dtype).contiguous()
    dx = torch.empty(n, c, dtype=x.dtype, device=x.device)
    dres = torch.empty(n, c, dtype=x.dtype, device=x.device) if (want_dres and y_mask is not None) else None
    dw = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _bn_workspace(x.device, c)
    N.check(N.lib().pbn_bn_act_train_backward(
        N.c_vp(x.data_ptr()), x.stride(0), N.c_vp(dy.data_ptr()), dy.stride(0),
        None if y_mask is None else N.c_vp(y_mask.data_ptr()), 0 if y_mask is None else y_mask.stride(0), n, c, _DT[x.dtype],
        N.ptr(weight), N.ptr(mean), N.ptr(invstd), N.c_vp(dx.data_ptr()), c, None if dres is None else N.c_vp(dres.data_ptr()), c,
        N.ptr(dw), N.ptr(db), N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream()), "pbn_bn_act_train_backward")
    if want_dres and dres is None:
        dres = dy
    return dx, dres, dw, db


def _conv_maps(conv, x):
    nbr, out_stride, dgrad_nbr, flip = conv._map(x)
    return nbr, out_stride, dgrad_nbr, flip, x.coordinate_manager.num_rows(out_stride)


def _conv_backward(conv, kernel, feats, g, nbr, dgrad_nbr, flip, want_dx, residual=None):
    """Input gradient (+ `residual` in the epilogue) and kernel gradient of one convolution; g = d(conv output)."""
    dx = None
    if want_dx:
        packed = conv._cache.get_dgrad(kernel, g.dtype, flip)
        dx = spconv_forward(g, dgrad_nbr, feats.shape[0], packed, residual=residual)
        if dx.shape[1] != feats.shape[1]:
            dx = dx[:, :feats.shape[1]]
    k3 = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
    dk = wgrad_native(feats, g, nbr, int(k3.shape[1]), int(k3.shape[2])).to(kernel.dtype).view_as(kernel)
    return dx, dk


class _ConvBnActFn(torch.autograd.Function):
    """relu(bn(conv(x))) -- the stem, the strided and the transposed convolutions of the U-Net (Mink.py:293-338)."""

    @staticmethod
    def forward(ctx, feats, kernel, bn_w, bn_b, conv, norm, nbr, dgrad_nbr, flip, n_out, relu):
        pre = spconv_forward(feats, nbr, n_out, conv._cache.get(kernel, feats.dtype))
        y, mean, invstd = _bn_forward(pre, norm, None, relu)
        ctx.save_for_backward(feats, kernel, bn_w, pre, y, mean, invstd)
        ctx.conv, ctx.nbr, ctx.dgrad_nbr, ctx.flip, ctx.relu = conv, nbr, dgrad_nbr, flip, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        feats, kernel, bn_w, pre, y, mean, invstd = ctx.saved_tensors
        g, _, dw, db = _bn_backward(pre, dy, y if ctx.relu else None, bn_w, mean, invstd, False)
        dx, dk = _conv_backward(ctx.conv, kernel, feats, g, ctx.nbr, ctx.dgrad_nbr, ctx.flip, ctx.needs_input_grad[0])
        return dx, dk, dw, db, None, None, None, None, None, None, None


def conv_bn_act(conv, norm, x, relu=True):
    """norm -> relu of `conv(x)` as one autograd node on the native training path (the three modules otherwise)."""
    from .nn import bn_act
    if conv.bias is None and usable(x.F, norm):
        nbr, out_stride, dgrad_nbr, flip, n_out = _conv_maps(conv, x)
        y = _ConvBnActFn.apply(x.F, conv.kernel, norm.bn.weight, norm.bn.bias, conv, norm, nbr, dgrad_nbr, flip, n_out, relu)
        from .core import SparseTensor
        return SparseTensor(y, coordinate_manager=x.coordinate_manager, tensor_stride=out_stride)
    return bn_act(norm, conv(x), relu=relu)


class _BasicBlockFn(torch.autograd.Function):
    """BasicBlock.forward (conv1 -> norm1 -> relu -> conv2 -> norm2 -> += residual | downsample(x) -> relu) in one node."""

    @staticmethod
    def forward(ctx, feats, k1, w1, b1, k2, w2, b2, kd, wd, bd, blk, nbr, n):
        c1, c2 = blk.conv1, blk.conv2
        h_pre = spconv_forward(feats, nbr, n, c1._cache.get(k1, feats.dtype))
        h, m1, s1 = _bn_forward(h_pre, blk.norm1, None, True)
        o_pre = spconv_forward(h, nbr, n, c2._cache.get(k2, feats.dtype))
        if kd is not None:
            r_pre = spconv_forward(feats, None, n, blk.downsample[0]._cache.get(kd, feats.dtype))
            res, md, sd = _bn_forward(r_pre, blk.downsample[1], None, False)
        else:
            r_pre = md = sd = None
            res = feats
        y, m2, s2 = _bn_forward(o_pre, blk.norm2, res, True)
        ctx.save_for_backward(feats, k1, w1, k2, w2, kd, wd, h_pre, h, o_pre, y, m1, s1, m2, s2, r_pre, md, sd)
        ctx.blk, ctx.nbr = blk, nbr
        return y

    @staticmethod
    def backward(ctx, dy):
        feats, k1, w1, k2, w2, kd, wd, h_pre, h, o_pre, y, m1, s1, m2, s2, r_pre, md, sd = ctx.saved_tensors
        blk, nbr = ctx.blk, ctx.nbr
        want_dx = ctx.needs_input_grad[0]
        g2, dres, dw2, db2 = _bn_backward(o_pre, dy, y, w2, m2, s2, want_dx or kd is not None)
        dh, dk2 = _conv_backward(blk.conv2, k2, h, g2, nbr, nbr, True, True)
        g1, _, dw1, db1 = _bn_backward(h_pre, dh, h, w1, m1, s1, False)
        dkd = dwd = dbd = None
        if kd is None:
            # the identity branch's gradient enters as the residual of conv1's input-gradient launch
            dx, dk1 = _conv_backward(blk.conv1, k1, feats, g1, nbr, nbr, True, want_dx, residual=dres if want_dx else None)
        else:
            dx, dk1 = _conv_backward(blk.conv1, k1, feats, g1, nbr, nbr, True, want_dx)
            gd, _, dwd, dbd = _bn_backward(r_pre, dres, None, wd, md, sd, False)
            dx, dkd = _conv_backward(blk.downsample[0], kd, feats, gd, None, None, False, want_dx, residual=dx)
        return dx, dk1, dw1, db1, dk2, dw2, db2, dkd, dwd, dbd, None, None, None


def basic_block(blk, x):
    """The block on the native training path; None when it does not apply (the caller then runs the modules)."""
    ds = blk.downsample
    norms = (blk.norm1, blk.norm2) + ((ds[1],) if ds is not None else ())
    convs = (blk.conv1, blk.conv2) + ((ds[0],) if ds is not None else ())
    if not usable(x.F, *norms) or any(c.bias is not None or c.stride != 1 or c.is_transpose for c in convs):
        return None
    if blk.conv1.kernel_size != 3 or blk.conv2.kernel_size != 3 or (ds is not None and ds[0].kernel_size != 1):
        return None
    nbr, out_stride, _, _, n = _conv_maps(blk.conv1, x)
    kd, wd, bd = (ds[0].kernel, ds[1].bn.weight, ds[1].bn.bias) if ds is not None else (None, None, None)
    y = _BasicBlockFn.apply(x.F, blk.conv1.kernel, blk.norm1.bn.weight, blk.norm1.bn.bias, blk.conv2.kernel,
                            blk.norm2.bn.weight, blk.norm2.bn.bias, kd, wd, bd, blk, nbr, n)
    return x.replace_feature(y)
