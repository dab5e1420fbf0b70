"""Sparse convolution modules over libpbnet_hip.so's implicit-GEMM kernel (csrc/spconv.hip).

Mirrors the constructor signatures and parameter names the reference uses from MinkowskiEngine
(/root/reference/network/Mink.py:221-288: MinkowskiConvolution / MinkowskiConvolutionTranspose with `.kernel`
[K^3,Cin,Cout] or [Cin,Cout] and optional `.bias` [1,Cout]; network/PBNet.py:43-82: MinkowskiLinear with `.linear`).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from ..scratch import StreamScratch
from .core import SparseTensor

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_ELEMS = {torch.float32: 4, torch.bfloat16: 8, torch.float16: 8}


def _vpo(cin, dtype):
    e = _ELEMS[dtype]
    v = (cin + e - 1) // e
    return v if v in (1, 2) else (v + 3) // 4 * 4


def pack_weight(kernel, dtype, flip=False, transpose=False):
    """kernel [K, A, B] (fp32 master) -> MFMA-fragment order [n_steps, cout_p/16, 64, 16 bytes] of `dtype`, where the
    convolution's (Cin, Cout) = (B, A) if transpose else (A, B) and flip mirrors the offsets (flip + transpose of a
    centred kernel = its input-gradient weights).  One native launch on the device (pbn_pack_weight).

    Layout contract: include/pbnet_hip.h (pbn_spconv_forward, `w_packed`)."""
    k = kernel.shape[0]
    cin, cout = (kernel.shape[2], kernel.shape[1]) if transpose else (kernel.shape[1], kernel.shape[2])
    e = _ELEMS[dtype]
    vpo = _vpo(cin, dtype)
    cin_p = vpo * e
    cout_p = (cout + 15) // 16 * 16
    n_steps = (k * vpo + 3) // 4
    chunk = 4 * e
    if kernel.is_cuda:
        src = kernel.detach()
        if src.dtype != torch.float32 or not src.is_contiguous():
            src = src.float().contiguous()
        w = torch.empty(n_steps, cout_p // 16, 64, e, dtype=dtype, device=kernel.device)
        N.check(N.lib().pbn_pack_weight(N.ptr(src), int(k), int(kernel.shape[1]), int(kernel.shape[2]), int(flip),
                                        int(transpose), _DT[dtype], vpo, n_steps, cout_p, N.c_vp(w.data_ptr()),
                                        N.current_stream()), "pbn_pack_weight")
        return w, vpo, n_steps, cout_p
    if flip:
        kernel = kernel.flip(0)
    if transpose:
        kernel = kernel.transpose(1, 2)
    w = torch.zeros(n_steps * chunk, cout_p, dtype=torch.float32, device=kernel.device)
    wp = torch.zeros(k, cin_p, cout_p, dtype=torch.float32, device=kernel.device)
    wp[:, :cin, :cout] = kernel.detach().float()
    w[:k * cin_p] = wp.reshape(k * cin_p, cout_p)
    # [S, g(4), j(E), T, c(16)] -> [S, T, g, c, j]
    w = w.reshape(n_steps, 4, e, cout_p // 16, 16).permute(0, 3, 1, 4, 2).contiguous().to(dtype)
    return w, vpo, n_steps, cout_p


class _PackCache(object):
    """Per-module cache of packed weights keyed by (dtype, parameter version)."""

    def __init__(self):
        self.store = {}

    def get(self, kernel, dtype):
        key = (dtype, kernel._version, kernel.data_ptr(), kernel.device)
        hit = self.store.get(dtype)
        if hit is None or hit[0] != key:
            k3 = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
            hit = (key, pack_weight(k3, dtype))
            self.store[dtype] = hit
        return hit[1]

    def get_linear(self, weight, dtype):
        """nn.Linear weight [out, in] -> packed [1, in, out]."""
        key = (dtype, weight._version, weight.data_ptr(), weight.device)
        hit = self.store.get(dtype)
        if hit is None or hit[0] != key:
            hit = (key, pack_weight(weight.detach().unsqueeze(0), dtype, transpose=True))
            self.store[dtype] = hit
        return hit[1]


SPLITK_WORKSPACE_BYTES = 256 << 20
_WS = StreamScratch()


def _workspace(device):
    """Scratch for split-K launches (fp32 partial slabs), one block per (device, stream): launches on one stream are
    ordered, two scenes in flight on two streams must not share the slabs."""
    return _WS.get(device, SPLITK_WORKSPACE_BYTES)


def spconv_forward(feats, nbr, n_out, packed, scale=None, shift=None, residual=None, relu=False, out=None,
                   row_perm=None, rows_per_wave=0):
    """Launch pbn_spconv_forward.  feats [n_in, C] (row stride = feats.stride(0)), nbr int32 [n_out, K] or None.

    Returns the output slab view [n_out, cout_p] (writes into `out` when given: a [n_out, >=cout_p] strided view)."""
    w, vpo, n_steps, cout_p = packed
    dtype = feats.dtype
    e = _ELEMS[dtype]
    cin_p = vpo * e
    assert feats.stride(1) == 1
    if feats.shape[1] < cin_p or (feats.stride(0) * feats.element_size()) % 16 or feats.data_ptr() % 16:
        padded = torch.zeros(feats.shape[0], cin_p, dtype=dtype, device=feats.device)
        padded[:, :feats.shape[1]] = feats
        feats = padded
    k = 1 if nbr is None else int(nbr.shape[1])
    assert n_steps == (k * vpo + 3) // 4, "packed weight does not match the kernel map"
    if out is None:
        out = torch.empty(n_out, cout_p, dtype=dtype, device=feats.device)
    assert out.stride(1) == 1 and out.shape[1] >= cout_p and out.dtype == dtype
    if residual is not None:
        assert residual.stride(1) == 1 and residual.dtype == dtype and residual.shape[1] >= cout_p
    ws = _workspace(feats.device)
    rc = N.lib().pbn_spconv_forward(
        N.c_vp(feats.data_ptr()), feats.stride(0), int(feats.shape[0]), None if nbr is None else N.c_vp(nbr.data_ptr()), k,
        None if row_perm is None else N.c_vp(row_perm.data_ptr()), None, int(n_out), N.c_vp(w.data_ptr()), vpo, n_steps,
        cout_p, None if scale is None else N.c_vp(scale.data_ptr()), None if shift is None else N.c_vp(shift.data_ptr()),
        None if residual is None else N.c_vp(residual.data_ptr()), 0 if residual is None else residual.stride(0),
        int(bool(relu)), N.c_vp(out.data_ptr()), out.stride(0), _DT[dtype], int(rows_per_wave),
        N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream())
    N.check(rc, "pbn_spconv_forward")
    return out


def _pad_vec(v, cout_p, fill):
    out = torch.full((cout_p,), fill, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v.reshape(-1).float()
    return out


WGRAD_PAIR_SEGMENT = 4096         # rule pairs per batched-GEMM segment of the compacted weight gradient


def _gather_rows(src, idx):
    """out[i] = src[idx[i]] (zero row for idx < 0), 16-byte rows, one launch (pbn_gather_rows)."""
    es = src.element_size()
    out = torch.empty(idx.shape[0], src.shape[1], dtype=src.dtype, device=src.device)
    N.check(N.lib().pbn_gather_rows(N.c_vp(src.data_ptr()), src.stride(0) * es, N.ptr(idx), int(idx.shape[0]),
                                    src.shape[1] * es, N.c_vp(out.data_ptr()), src.shape[1] * es, N.current_stream()),
            "pbn_gather_rows")
    return out


def rulebook_pairs(nbr, segment=WGRAD_PAIR_SEGMENT):
    """The map as offset-major lists of (input row, output row) pairs cut into segments of `segment` pairs, -1 padded:
    (in_idx, out_idx) int64 [segments * segment], seg_offset int64 [segments], segments.  One read-back (pairs per
    offset) sizes the lists.  Cached on the map tensor: every layer of a level shares its map."""
    hit = getattr(nbr, "_pbn_pairs", None)
    if hit is not None and hit[4] == segment:
        return hit[:4]
    N.require_cuda(nbr)
    assert nbr.dtype == torch.int32 and nbr.is_contiguous()
    lib = N.lib()
    v, k = int(nbr.shape[0]), int(nbr.shape[1])
    dev = nbr.device
    table = torch.empty(max(lib.pbn_rulebook_pair_blocks(v), 1) * k, dtype=torch.int32, device=dev)
    totals = torch.empty(k, dtype=torch.int32, device=dev)
    st = N.current_stream()
    N.check(lib.pbn_rulebook_pair_counts(N.ptr(nbr), v, k, N.ptr(table), N.ptr(totals), st), "pbn_rulebook_pair_counts")
    cnt = totals.cpu().numpy().astype(np.int64)                                  # the one read-back
    segs = (cnt + segment - 1) // segment
    seg_begin = np.concatenate([[0], np.cumsum(segs)]).astype(np.int32)        # [K+1]
    seg_start = seg_begin[:-1]
    n_seg = int(segs.sum())
    in_idx = torch.empty(n_seg * segment, dtype=torch.int64, device=dev)
    out_idx = torch.empty(n_seg * segment, dtype=torch.int64, device=dev)
    seg_offset = torch.empty(n_seg, dtype=torch.int64, device=dev)
    seg_begin_d = torch.from_numpy(seg_begin).to(dev)
    N.check(lib.pbn_rulebook_pair_fill(N.ptr(nbr), v, k, N.ptr(table), N.ptr(seg_begin_d), segment,
                                       n_seg, N.ptr(in_idx), N.ptr(out_idx), N.ptr(seg_offset), st), "pbn_rulebook_pair_fill")
    hit = (in_idx, out_idx, seg_offset, n_seg, segment, seg_begin_d)
    try:
        nbr._pbn_pairs = hit
    except AttributeError:
        pass
    return hit[:4]


_WGRAD_WS = StreamScratch()


def wgrad_native(feats, grad_out, nbr, cin, cout):
    """dW[k] = sum over the rule pairs (i, o) of offset k of feats[i]^T grad_out[o] on the matrix cores
    (pbn_spconv_wgrad, csrc/wgrad.hip): pair lists from pbn_rulebook_pair_* (cached per map; every layer of a level shares
    them), operands read in place (no gathered copies), fixed summation order, fp32 result [K, cin, cout].
    nbr None = identity pairs (1x1 convolution / linear layer)."""
    N.require_cuda(feats, grad_out)
    assert feats.dtype == grad_out.dtype and feats.stride(1) == 1 and grad_out.stride(1) == 1
    dev = feats.device
    lib = N.lib()
    if nbr is None:
        k, in_idx, out_idx, seg_begin, n_pairs = 1, None, None, None, int(feats.shape[0])
    else:
        assert nbr.is_contiguous()
        k = int(nbr.shape[1])
        rulebook_pairs(nbr)
        in_idx, out_idx, _, n_seg, segment, seg_begin = nbr._pbn_pairs
        n_pairs = n_seg * segment
    dw = torch.empty(k, cin, cout, dtype=torch.float32, device=dev)
    ws = _WGRAD_WS.get(dev, int(lib.pbn_spconv_wgrad_workspace_bytes(k, cin, cout)))
    rc = lib.pbn_spconv_wgrad(N.c_vp(feats.data_ptr()), feats.stride(0), N.c_vp(grad_out.data_ptr()), grad_out.stride(0),
                              _DT[feats.dtype], N.ptr(in_idx), N.ptr(out_idx), N.ptr(seg_begin),
                              WGRAD_PAIR_SEGMENT if nbr is not None else 0, n_pairs, k, int(cin), int(cout), N.ptr(dw),
                              N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream())
    N.check(rc, "pbn_spconv_wgrad")
    return dw


def _wgrad(feats, grad_out, nbr, cin, cout):
    return wgrad_native(feats, grad_out, nbr, cin, cout)


class _ConvFn(torch.autograd.Function):
    """Sparse convolution with autograd.  forward / dgrad run on the implicit-GEMM kernel (the input gradient of an
    output-stationary table is the same kind of table: the mirrored offset of the same map for odd kernels, the up table
    for a k=2,s=2 convolution, the down table for its transpose); wgrad is a gather + library GEMM (see _wgrad)."""

    @staticmethod
    def forward(ctx, feats, kernel, bias, nbr, n_out, cache, dgrad_nbr, flip):
        packed = cache.get(kernel, feats.dtype)
        shift = None if bias is None else _pad_vec(bias, packed[3], 0.0)
        out = spconv_forward(feats, nbr, n_out, packed, shift=shift)
        cout = kernel.shape[-1]
        ctx.save_for_backward(feats, kernel)
        ctx.nbr, ctx.dgrad_nbr, ctx.flip, ctx.has_bias = nbr, dgrad_nbr, flip, bias is not None
        return out if out.shape[1] == cout else out[:, :cout]

    @staticmethod
    def backward(ctx, grad_out):
        feats, kernel = ctx.saved_tensors
        k3 = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
        cin, cout = k3.shape[1], k3.shape[2]
        grad_out = grad_out.contiguous()
        grad_feats = grad_kernel = grad_bias = None
        if ctx.needs_input_grad[0]:
            packed = pack_weight(k3.detach(), grad_out.dtype, flip=bool(ctx.flip), transpose=True)   # [K, Cout, Cin]
            gi = spconv_forward(grad_out, ctx.dgrad_nbr, feats.shape[0], packed)
            grad_feats = gi if gi.shape[1] == feats.shape[1] else gi[:, :feats.shape[1]]
        if ctx.needs_input_grad[1]:
            grad_kernel = _wgrad(feats, grad_out, ctx.nbr, cin, cout).to(kernel.dtype).view_as(kernel)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = grad_out.float().sum(0, keepdim=True)
        return grad_feats, grad_kernel, grad_bias, None, None, None, None, None


class MinkowskiConvolutionBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, is_transpose=False, dimension=None):
        super().__init__()
        assert dimension in (None, 3), "only 3-D sparse tensors are on the path"
        assert dilation == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.is_transpose = int(kernel_size), int(stride), is_transpose
        self.kernel_volume = self.kernel_size ** 3
        self.use_mm = (self.kernel_volume == 1 and self.stride == 1)  # ME: plain matmul, kernel [Cin, Cout]
        shape = (in_channels, out_channels) if self.use_mm else (self.kernel_volume, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self._cache = _PackCache()
        self.reset_parameters()

    def reset_parameters(self):
        # ME's default: uniform(-1/sqrt(n), 1/sqrt(n)), n = (out if transposed else in) * kernel_volume
        with torch.no_grad():
            n = (self.out_channels if self.is_transpose else self.in_channels) * self.kernel_volume
            stdv = 1.0 / math.sqrt(n)
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def _map(self, x):
        """(forward table, output stride, table of the input gradient, mirror-the-offsets flag)."""
        cm, s = x.coordinate_manager, x.tensor_stride
        if self.is_transpose:
            assert self.kernel_size == 2 and self.stride == 2, "only k=2,s=2 transposed convolutions are on the path"
            return cm.up_map(s), s // 2, cm.down_map(s // 2), False
        if self.kernel_size == 1 and self.stride == 1:
            return None, s, None, False
        if self.stride == 2:
            assert self.kernel_size == 2, "only k=2,s=2 strided convolutions are on the path"
            return cm.down_map(s), s * 2, cm.up_map(s * 2), False
        assert self.stride == 1 and self.kernel_size % 2 == 1
        nbr = cm.kernel_map(s, self.kernel_size)
        return nbr, s, nbr, True   # centred cube: offset K-1-k is the mirror of offset k, same table serves dgrad

    def forward(self, x):
        nbr, out_stride, dgrad_nbr, flip = self._map(x)
        n_out = x.coordinate_manager.num_rows(out_stride)
        feats = _ConvFn.apply(x.F, self.kernel, self.bias, nbr, n_out, self._cache, dgrad_nbr, flip)
        return SparseTensor(feats, coordinate_manager=x.coordinate_manager, tensor_stride=out_stride)

    def extra_repr(self):
        return "in=%d, out=%d, kernel_size=%d, stride=%d%s" % (self.in_channels, self.out_channels, self.kernel_size,
                                                                self.stride, ", transposed" if self.is_transpose else "")


class MinkowskiConvolution(MinkowskiConvolutionBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator, False,
                         dimension)


class MinkowskiConvolutionTranspose(MinkowskiConvolutionBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator, True,
                         dimension)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, weight, bias, cache):
        packed = cache.get_linear(weight, feats.dtype)
        shift = None if bias is None else _pad_vec(bias, packed[3], 0.0)
        out = spconv_forward(feats, None, feats.shape[0], packed, shift=shift)
        cout = weight.shape[0]
        ctx.save_for_backward(feats, weight)
        ctx.has_bias = bias is not None
        return out if out.shape[1] == cout else out[:, :cout]

    @staticmethod
    def backward(ctx, grad_out):
        feats, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_feats = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            packed = pack_weight(weight.detach().unsqueeze(0), grad_out.dtype)      # [1, out, in]: grad_x = grad_out @ W
            gi = spconv_forward(grad_out, None, feats.shape[0], packed)
            grad_feats = gi if gi.shape[1] == feats.shape[1] else gi[:, :feats.shape[1]]
        if ctx.needs_input_grad[1]:                      # dW[out, in] = (x^T g)^T on the same pair-contraction kernel
            grad_weight = wgrad_native(feats, grad_out, None, feats.shape[1], grad_out.shape[1])[0].t().to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = grad_out.float().sum(0).to(weight.dtype)
        return grad_feats, grad_weight, grad_bias, None


class MinkowskiLinear(nn.Module):
    """ME.MinkowskiLinear: `self.linear = nn.Linear` on the feature matrix (state-dict key `linear.weight`)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        self._cache = _PackCache()

    def forward(self, x):
        feats = _LinearFn.apply(x.F, self.linear.weight, self.linear.bias, self._cache)
        return x.replace_feature(feats)
