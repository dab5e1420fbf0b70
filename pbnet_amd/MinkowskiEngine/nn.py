"""Feature-wise modules of the ME subset (/root/reference/network/PBNet.py:43-85, network/Mink.py:224,288):
thin wrappers that apply a torch module to the feature matrix, exactly as MinkowskiEngine does."""
import torch
import torch.nn as nn

from .core import SparseTensor


class MinkowskiBatchNorm(nn.Module):
    """ME.MinkowskiBatchNorm: `self.bn = nn.BatchNorm1d` on .F (Mink.py:71-73 reaches into `.bn`)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        # torch's batch-norm kernels take bf16/f16 slabs with fp32 parameters and statistics directly (identical
        # output to an fp32 round trip, two conversion launches fewer in each direction)
        return x.replace_feature(self.bn(x.F))


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
