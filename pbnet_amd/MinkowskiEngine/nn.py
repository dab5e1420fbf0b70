"""Feature-wise modules of the ME subset (/root/reference/network/PBNet.py:43-85, network/Mink.py:224,288):
thin wrappers that apply a torch module to the feature matrix, exactly as MinkowskiEngine does."""
import torch
import torch.nn as nn

from ..scratch import StreamScratch
from .core import SparseTensor


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_BN_WS = StreamScratch()


def _bn_workspace(device, channels):
    """Per (device, stream) scratch of the native batch norm (block partial sums), grown on demand; new blocks are zeroed (the
    ticket counter of the fused merge step, include/pbnet_hip.h)."""
    from .. import _native as N
    return _BN_WS.get(device, int(N.lib().pbn_bn_workspace_bytes(int(channels))), min_bytes=1 << 20, zero=True)


def _rows_ok(t):
    es = t.element_size()
    return t.dim() == 2 and t.stride(1) == 1 and (t.stride(0) * es) % 16 == 0 and t.data_ptr() % 16 == 0 \
        and (t.shape[1] * es) % 16 == 0 and t.shape[0] > 0


class _BatchNormTrainFn(torch.autograd.Function):
    """Train-mode nn.BatchNorm1d on a feature slab through pbn_bn_train_forward / _backward (csrc/bnorm.hip): same
    statistics (biased variance to normalise, unbiased into running_var), fp32 arithmetic on f32 / bf16 / f16 slabs."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum):
        from .. import _native as N
        n, c = int(x.shape[0]), int(x.shape[1])
        y = torch.empty(n, c, dtype=x.dtype, device=x.device)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        ws = _bn_workspace(x.device, c)
        N.check(N.lib().pbn_bn_train_forward(N.c_vp(x.data_ptr()), x.stride(0), n, c, _DT[x.dtype], N.ptr(weight), N.ptr(bias),
                                             float(eps), float(momentum), N.ptr(running_mean), N.ptr(running_var),
                                             N.c_vp(y.data_ptr()), c, N.ptr(mean), N.ptr(invstd), N.c_vp(ws.data_ptr()),
                                             ws.numel(), N.current_stream()), "pbn_bn_train_forward")
        ctx.save_for_backward(x, weight, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _native as N
        x, weight, mean, invstd = ctx.saved_tensors
        n, c = int(x.shape[0]), int(x.shape[1])
        if dy.dtype != x.dtype or not _rows_ok(dy):
            dy = dy.to(x.dtype).contiguous()
        dx = torch.empty(n, c, dtype=x.dtype, device=x.device)
        need_w = weight is not None and ctx.needs_input_grad[1]
        dw = torch.empty(c, dtype=torch.float32, device=x.device) if need_w else None
        db = torch.empty(c, dtype=torch.float32, device=x.device) if (weight is not None and ctx.needs_input_grad[2]) else None
        ws = _bn_workspace(x.device, c)
        N.check(N.lib().pbn_bn_train_backward(N.c_vp(x.data_ptr()), x.stride(0), N.c_vp(dy.data_ptr()), dy.stride(0), n, c,
                                              _DT[x.dtype], N.ptr(weight), N.ptr(mean), N.ptr(invstd), N.c_vp(dx.data_ptr()), c,
                                              N.ptr(dw), N.ptr(db), N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream()),
                "pbn_bn_train_backward")
        return dx, dw, db, None, None, None, None


class _BatchNormActTrainFn(torch.autograd.Function):
    """Train-mode batch norm with the tail of the reference's blocks fused in (Mink.py:293-350: conv -> bn -> relu,
    conv -> bn -> += residual -> relu): y = relu(bn(x) [+ residual]) in the normalisation's apply pass; the backward masks
    dy with (y > 0) while it reads it and hands the masked gradient to the residual branch (pbn_bn_act_train_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, eps, momentum, relu):
        from .. import _native as N
        n, c = int(x.shape[0]), int(x.shape[1])
        y = torch.empty(n, c, dtype=x.dtype, device=x.device)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        ws = _bn_workspace(x.device, c)
        N.check(N.lib().pbn_bn_act_train_forward(
            N.c_vp(x.data_ptr()), x.stride(0), n, c, _DT[x.dtype], N.ptr(weight), N.ptr(bias), float(eps), float(momentum),
            N.ptr(running_mean), N.ptr(running_var), None if residual is None else N.c_vp(residual.data_ptr()),
            0 if residual is None else residual.stride(0), int(bool(relu)), N.c_vp(y.data_ptr()), c, N.ptr(mean), N.ptr(invstd),
            N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream()), "pbn_bn_act_train_forward")
        ctx.save_for_backward(x, weight, mean, invstd, y if relu else None)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _native as N
        x, weight, mean, invstd, y = ctx.saved_tensors
        n, c = int(x.shape[0]), int(x.shape[1])
        if dy.dtype != x.dtype or not _rows_ok(dy):
            dy = dy.to(x.dtype).contiguous()
        dx = torch.empty(n, c, dtype=x.dtype, device=x.device)
        want_res = ctx.has_res and ctx.needs_input_grad[3]
        # without an activation the residual branch receives dy itself: no copy
        dres = torch.empty(n, c, dtype=x.dtype, device=x.device) if (want_res and y is not None) else None
        need_w = weight is not None and ctx.needs_input_grad[1]
        dw = torch.empty(c, dtype=torch.float32, device=x.device) if need_w else None
        db = torch.empty(c, dtype=torch.float32, device=x.device) if (weight is not None and ctx.needs_input_grad[2]) else None
        ws = _bn_workspace(x.device, c)
        N.check(N.lib().pbn_bn_act_train_backward(
            N.c_vp(x.data_ptr()), x.stride(0), N.c_vp(dy.data_ptr()), dy.stride(0), None if y is None else N.c_vp(y.data_ptr()),
            0 if y is None else y.stride(0), n, c, _DT[x.dtype], N.ptr(weight), N.ptr(mean), N.ptr(invstd), N.c_vp(dx.data_ptr()), c,
            None if dres is None else N.c_vp(dres.data_ptr()), c, N.ptr(dw), N.ptr(db), N.c_vp(ws.data_ptr()), ws.numel(),
            N.current_stream()), "pbn_bn_act_train_backward")
        if want_res and dres is None:
            dres = dy
        return dx, dw, db, dres, None, None, None, None, None


def bn_act(norm, x, residual=None, relu=True):
    """relu(norm(x) [+ residual]) of the reference's blocks as ONE pass each way when `norm` (a MinkowskiBatchNorm) is
    training on the native path; the separate modules otherwise (same values: norm -> += residual -> relu)."""
    f, bn = x.F, norm.bn
    res = None if residual is None else residual.F
    if norm.NATIVE_TRAIN and norm.FUSE_ACT and bn.training and f.is_cuda and bn.momentum is not None and f.dtype in _DT \
            and _rows_ok(f) and (bn.weight is None or bn.weight.dtype == torch.float32) \
            and (res is None or (res.dtype == f.dtype and _rows_ok(res) and res.shape == f.shape)):
        norm._tick()
        rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
        return x.replace_feature(_BatchNormActTrainFn.apply(f, bn.weight, bn.bias, res, rm, rv, bn.eps, bn.momentum, relu))
    out = norm(x)
    if residual is not None:
        out = out.replace_feature(out.F + residual.F)
    return out.replace_feature(torch.relu(out.F)) if relu else out


def _flush_ticks_hook(module, prefix, keep_vars):
    module.flush_ticks()


def _drop_ticks_hook(module, state_dict, prefix, *unused):
    if prefix + "num_batches_tracked" in state_dict:
        module.__dict__["_pending_ticks"] = 0


class _TickedBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d whose num_batches_tracked increments are counted on the host and applied when somebody looks at the
    buffer (attribute access, state_dict): one tiny launch per layer and step less on the native training path."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__["_pending_ticks"] = 0
        self.register_state_dict_pre_hook(_flush_ticks_hook)            # a module-level function: the module stays picklable
        self.register_load_state_dict_pre_hook(_drop_ticks_hook)        # a loaded counter replaces the pending increments

    def flush_ticks(self):
        t = self.__dict__.get("_pending_ticks", 0)
        if t:
            self.__dict__["_pending_ticks"] = 0
            buf = self._buffers.get("num_batches_tracked")
            if buf is not None:
                buf.add_(t)

    def __getattr__(self, name):
        if name == "num_batches_tracked":
            self.flush_ticks()
        return super().__getattr__(name)

    def _apply(self, fn, *a, **k):            # .to() / .half() / .cuda() walk _buffers directly
        self.flush_ticks()
        return super()._apply(fn, *a, **k)

    def buffers(self, *a, **k):
        self.flush_ticks()
        return super().buffers(*a, **k)

    def named_buffers(self, *a, **k):
        self.flush_ticks()
        return super().named_buffers(*a, **k)


class MinkowskiBatchNorm(nn.Module):
    """ME.MinkowskiBatchNorm: `self.bn = nn.BatchNorm1d` on .F (Mink.py:71-73 reaches into `.bn`)."""

    NATIVE_TRAIN = True          # train-mode statistics / normalisation / gradients through csrc/bnorm.hip
    FUSE_ACT = True              # bn_act(): residual add + ReLU inside the normalisation passes

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = _TickedBatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                     track_running_stats=track_running_stats)

    def _tick(self):
        bn = self.bn
        if bn.track_running_stats and bn._buffers.get("num_batches_tracked") is not None:
            bn.__dict__["_pending_ticks"] = bn.__dict__.get("_pending_ticks", 0) + 1

    def forward(self, x):
        f, bn = x.F, self.bn
        if self.NATIVE_TRAIN and bn.training and f.is_cuda and bn.momentum is not None and f.dtype in _DT and _rows_ok(f) \
                and (bn.weight is None or bn.weight.dtype == torch.float32):
            self._tick()
            rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
            return x.replace_feature(_BatchNormTrainFn.apply(f, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum))
        # torch's batch-norm kernels take bf16/f16 slabs with fp32 parameters and statistics directly (identical
        # output to an fp32 round trip, two conversion launches fewer in each direction)
        return x.replace_feature(bn(f))


class _Elementwise(nn.Module):
    MODULE = None

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.module = self.MODULE(*args, **kwargs)

    def forward(self, x):
        f = x.F
        if f.dtype != torch.float32 and any(True for _ in self.module.parameters()):
            return x.replace_feature(self.module(f.float()).to(f.dtype))  # fp32 parameters (PReLU slope)
        return x.replace_feature(self.module(f))


class MinkowskiReLU(_Elementwise):
    MODULE = nn.ReLU


class MinkowskiPReLU(_Elementwise):
    MODULE = nn.PReLU


class MinkowskiSigmoid(_Elementwise):
    MODULE = nn.Sigmoid


class MinkowskiSoftmax(_Elementwise):
    MODULE = nn.Softmax

    def __init__(self, dim=1):
        super().__init__(dim=dim)


def _batch_index(x):
    assert x.tensor_stride >= 1
    b = x.C[:, 0].long()
    return b, int(b.max().item()) + 1 if b.numel() else 0


def segment_pool(feats, batch_sorted, n_batch, want_max=True, want_avg=True):
    """Global max / avg pooling per batch index for rows GROUPED by ascending batch index (pbn_segment_pool).
    Returns fp32 [n_batch, C] tensors (None for the one not requested)."""
    from .. import _native as N
    from .conv import _DT
    assert feats.stride(1) == 1
    seg_start = torch.searchsorted(batch_sorted.to(torch.int32).contiguous(),
                                   torch.arange(n_batch + 1, dtype=torch.int32, device=feats.device)).to(torch.int32)
    c = feats.shape[1]
    mx = torch.empty(n_batch, c, dtype=torch.float32, device=feats.device) if want_max else None
    av = torch.empty(n_batch, c, dtype=torch.float32, device=feats.device) if want_avg else None
    lib = N.lib()
    ws_bytes = int(lib.pbn_segment_pool_workspace_bytes(int(n_batch), c))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=feats.device)
    rc = lib.pbn_segment_pool(N.c_vp(feats.data_ptr()), feats.stride(0), c, _DT[feats.dtype], N.ptr(seg_start),
                              int(n_batch), N.ptr(mx), N.ptr(av), N.c_vp(ws.data_ptr()), ws_bytes, N.current_stream())
    N.check(rc, "pbn_segment_pool")
    return mx, av


class _SegmentPoolFn(torch.autograd.Function):
    """global max pooling + global average pooling per batch index (PBNet.py:274-276 adds the two) for rows grouped by
    ascending batch index: ONE deterministic segment-pool launch forward instead of scatter_reduce(amax) + index_add
    (1.3 ms of atomics on a ScanNet-sized proposal set); backward = the reductions' own rules: the average spreads its
    gradient evenly over the segment, the maximum evenly over the rows that attain it."""

    @staticmethod
    def forward(ctx, feats, batch_sorted, n_batch):
        mx, av = segment_pool(feats.detach(), batch_sorted, n_batch)
        b = batch_sorted.long()
        ctx.save_for_backward(feats, b, mx)
        ctx.n_batch = n_batch
        return mx + av

    @staticmethod
    def backward(ctx, g):
        feats, b, mx = ctx.saved_tensors
        g = g.float()
        f = feats.float()
        cnt = torch.bincount(b, minlength=ctx.n_batch).to(torch.float32).clamp_(min=1.0)
        gb = g[b]
        ties = (f == mx[b]).to(torch.float32)
        tie_cnt = torch.zeros(ctx.n_batch, f.shape[1], dtype=torch.float32, device=f.device).index_add_(0, b, ties)
        grad = gb / cnt[b][:, None] + gb * ties / tie_cnt[b].clamp_(min=1.0)
        return grad.to(feats.dtype), None, None


def global_max_plus_avg_pool(x):
    """MinkowskiGlobalMaxPooling()(x) + MinkowskiGlobalAvgPooling()(x) with autograd, rows grouped by batch index."""
    b, nb = _batch_index(x)
    return _PooledTensor(_SegmentPoolFn.apply(x.F, x.C[:, 0].contiguous(), nb).to(x.F.dtype))


class _GlobalPool(nn.Module):
    MODE = "avg"

    def forward(self, x):
        b, nb = _batch_index(x)
        f = x.F.float()
        if self.MODE == "avg":
            s = torch.zeros(nb, f.shape[1], dtype=f.dtype, device=f.device).index_add_(0, b, f)
            cnt = torch.zeros(nb, dtype=f.dtype, device=f.device).index_add_(0, b, torch.ones_like(b, dtype=f.dtype))
            out = s / cnt[:, None]
        else:
            out = torch.full((nb, f.shape[1]), float("-inf"), dtype=f.dtype, device=f.device)
            out = out.scatter_reduce(0, b[:, None].expand_as(f), f, reduce="amax", include_self=True)
        return _PooledTensor(out.to(x.F.dtype))


class _PooledTensor(object):
    """Result of a global pooling: one feature row per batch index, rows in ascending batch order."""

    def __init__(self, feats):
        self._F = feats

    @property
    def F(self):
        return self._F

    def replace_feature(self, feats):
        return _PooledTensor(feats)

    def __add__(self, other):
        return _PooledTensor(self._F + other._F)


class MinkowskiGlobalAvgPooling(_GlobalPool):
    MODE = "avg"


class MinkowskiGlobalMaxPooling(_GlobalPool):
    MODE = "max"
