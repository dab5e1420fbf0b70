"""Sparse convolution modules over libpbnet_hip.so's implicit-GEMM kernel (csrc/spconv.hip).

Mirrors the constructor signatures and parameter names the reference uses from MinkowskiEngine
(/root/reference/network/Mink.py:221-288: MinkowskiConvolution / MinkowskiConvolutionTranspose with `.kernel`
[K^3,Cin,Cout] or [Cin,Cout] and optional `.bias` [1,Cout]; network/PBNet.py:43-82: MinkowskiLinear with `.linear`).
"""
import math
import os
import threading
import weakref

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


_JOB = np.dtype([("src", "<u8"), ("out", "<u8"), ("n_offsets", "<i4"), ("dim_a", "<i4"), ("dim_b", "<i4"), ("flip", "<i4"),
                 ("transpose", "<i4"), ("cin", "<i4"), ("cout", "<i4"), ("cin_p", "<i4"), ("cout_p", "<i4"), ("n_steps", "<i4"),
                 ("reserved", "<i4", (2,))])          # include/pbnet_hip.h: pbn_pack_job (64 bytes)
_PACK_MODULES = weakref.WeakSet()                      # every live convolution module (MinkowskiConvolutionBase)


def _pack_dims(k, a, b, dtype, transpose):
    cin, cout = (b, a) if transpose else (a, b)
    e = _ELEMS[dtype]
    vpo = _vpo(cin, dtype)
    cout_p = (cout + 15) // 16 * 16
    return cin, cout, vpo, vpo * e, cout_p, (k * vpo + 3) // 4


class _BatchPacker(object):
    """Weight packing for ALL convolution layers in one launch (pbn_pack_weights_batch): a training step changes every
    kernel, so the first layer that finds its packed weights stale repacks the forward form of every stale layer on
    its device -- and, with autograd on, the input-gradient form (mirrored offsets, swapped channel roles) as well.
    Output buffers are reused while their size is unchanged, the device job table while the set of jobs is."""

    def __init__(self):
        self.lock = threading.Lock()
        self.tables = {}          # (device, dtype) -> (signature, device table, max_vectors, jobs)

    def refresh(self, device, dtype, with_dgrad):
        with self.lock:
            stream_raw = torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch._C._cuda_getDevice())
            jobs = []
            for m in list(_PACK_MODULES):
                w = m.kernel
                if w.device != device or w.dtype != torch.float32 or not w.is_contiguous():
                    continue
                key = (dtype, w._version, w.data_ptr(), w.device)
                forms = (("f", False, False), ("d", bool(m._dgrad_flip), True)) if (with_dgrad and w.requires_grad) else \
                    (("f", False, False),)
                for form, flip, transpose in forms:
                    hit = m._cache.store.get((form, dtype))
                    if hit is not None and hit[0] == key:
                        continue
                    k3 = w if w.dim() == 3 else w.unsqueeze(0)
                    k, a, b = int(k3.shape[0]), int(k3.shape[1]), int(k3.shape[2])
                    cin, cout, vpo, cin_p, cout_p, n_steps = _pack_dims(k, a, b, dtype, transpose)
                    # repack IN PLACE only what no other stream has fetched since it was packed: a convolution of another
                    # scene in flight may still be reading it (such readers were recorded on the buffer, see _PackCache:
                    # dropping it here is then safe, the allocator holds it back until their work has drained)
                    old = None if (hit is None or hit[2] != stream_raw or hit[3]) else hit[1][0]
                    shape = (n_steps, cout_p // 16, 64, _ELEMS[dtype])
                    out = old if (old is not None and tuple(old.shape) == shape and old.dtype == dtype) else \
                        torch.empty(*shape, dtype=dtype, device=device)
                    jobs.append((m, form, key, (out, vpo, n_steps, cout_p),
                                 (w.data_ptr(), out.data_ptr(), k, a, b, int(flip), int(transpose), cin, cout, cin_p, cout_p, n_steps)))
            if not jobs:
                return
            sig = tuple(j[4] for j in jobs)
            cached = self.tables.get((device, dtype))
            if cached is None or cached[0] != sig:
                host = np.zeros(len(jobs), dtype=_JOB)
                for i, j in enumerate(jobs):
                    host[i] = j[4] + ((0, 0),)
                table = torch.from_numpy(host.view(np.uint8).reshape(-1)).to(device)
                max_vec = max(j[4][11] * (j[4][10] // 16) * 64 for j in jobs)
                cached = (sig, table, max_vec)
                self.tables[(device, dtype)] = cached
            cached[1].record_stream(torch.cuda.current_stream(device))     # the table may have been built on another stream
            N.check(N.lib().pbn_pack_weights_batch(N.c_vp(cached[1].data_ptr()), len(jobs), int(cached[2]), _DT[dtype],
                                                   N.current_stream()), "pbn_pack_weights_batch")
            for m, form, key, packed, _ in jobs:
                m._cache.store[(form, dtype)] = (key, packed, stream_raw, set())


_BATCH = _BatchPacker()


class _PackCache(object):
    """Per-module cache of packed weights keyed by (form, dtype) -> (parameter version key, packed, raw stream of the pack
    launch, raw streams other than that one that have fetched the buffer).  A fetch from another stream is recorded on the
    buffer (Tensor.record_stream), so that a later repack or release cannot hand its memory out while that stream's
    convolutions still read it."""

    @staticmethod
    def _note_stream(hit):
        w = hit[1][0]
        if w.is_cuda:
            raw = torch._C._cuda_getCurrentRawStream(w.device.index)
            if raw != hit[2] and raw not in hit[3]:
                w.record_stream(torch.cuda.current_stream(w.device))
                hit[3].add(raw)
        return hit[1]

    def __init__(self, owner=None):
        self.store = {}
        self.owner = None if owner is None else weakref.ref(owner)

    def _lookup(self, form, kernel, dtype, flip, transpose):
        key = (dtype, kernel._version, kernel.data_ptr(), kernel.device)
        hit = self.store.get((form, dtype))
        if hit is not None and hit[0] == key:
            return self._note_stream(hit)
        owner = None if self.owner is None else self.owner()
        if owner is not None and kernel.is_cuda and kernel.data_ptr() == owner.kernel.data_ptr():
            _BATCH.refresh(kernel.device, dtype, torch.is_grad_enabled() or form == "d")
            hit = self.store.get((form, dtype))
            if hit is not None and hit[0] == key:
                return self._note_stream(hit)
        k3 = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
        hit = self._entry(key, pack_weight(k3.detach(), dtype, flip=flip, transpose=transpose))
        self.store[(form, dtype)] = hit
        return hit[1]

    @staticmethod
    def _entry(key, packed):
        w = packed[0]
        raw = torch._C._cuda_getCurrentRawStream(w.device.index) if w.is_cuda else 0
        return (key, packed, raw, set())

    def get(self, kernel, dtype):
        return self._lookup("f", kernel, dtype, False, False)

    def get_dgrad(self, kernel, dtype, flip):
        """Input-gradient weights of a forward kernel: offsets mirrored when `flip`, channel roles swapped."""
        return self._lookup("d", kernel, dtype, bool(flip), True)

    def get_linear(self, weight, dtype):
        """nn.Linear weight [out, in] -> packed [1, in, out]."""
        key = (dtype, weight._version, weight.data_ptr(), weight.device)
        hit = self.store.get(("l", dtype))
        if hit is None or hit[0] != key:
            hit = self._entry(key, pack_weight(weight.detach().unsqueeze(0), dtype, transpose=True))
            self.store[("l", dtype)] = hit
            return hit[1]
        return self._note_stream(hit)


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


def spconv_forward_dual(feats, nbr, n_out, feats2, packed, vpo2, shift=None, scale=None, relu=False, out=None, rows_per_wave=0):
    """pbn_spconv_forward_dual: the convolution over `nbr` plus a 1x1 over `feats2` (row o with output row o) in one reduction.
    packed = (w [steps of the map | steps of the second source (| zero padding)], vpo, n_steps_total, cout_p)."""
    w, vpo, n_steps, cout_p = packed
    dtype = feats.dtype
    assert feats.stride(1) == 1 and feats2.stride(1) == 1 and feats2.dtype == dtype
    if out is None:
        out = torch.empty(n_out, cout_p, dtype=dtype, device=feats.device)
    ws = _workspace(feats.device)
    rc = N.lib().pbn_spconv_forward_dual(
        N.c_vp(feats.data_ptr()), feats.stride(0), int(feats.shape[0]), N.c_vp(nbr.data_ptr()), int(nbr.shape[1]), None, int(n_out),
        N.c_vp(w.data_ptr()), vpo, n_steps, cout_p, None if scale is None else N.c_vp(scale.data_ptr()),
        None if shift is None else N.c_vp(shift.data_ptr()), None, 0, int(bool(relu)), N.c_vp(out.data_ptr()), out.stride(0),
        _DT[dtype], int(rows_per_wave), N.c_vp(ws.data_ptr()), ws.numel(), N.c_vp(feats2.data_ptr()), feats2.stride(0),
        int(feats2.shape[0]), int(vpo2), N.current_stream())
    N.check(rc, "pbn_spconv_forward_dual")
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
    (in_idx, out_idx) int32 [segments * segment], seg_offset int64 [segments], segments.  One read-back (pairs per
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
    in_idx = torch.empty(n_seg * segment, dtype=torch.int32, device=dev)
    out_idx = torch.empty(n_seg * segment, dtype=torch.int32, device=dev)
    seg_offset = torch.empty(n_seg, dtype=torch.int64, device=dev)
    seg_begin_d = torch.from_numpy(seg_begin).to(dev)
    N.check(lib.pbn_rulebook_pair_fill(N.ptr(nbr), v, k, N.ptr(table), N.ptr(seg_begin_d), segment,
                                       n_seg, N.ptr(in_idx), N.ptr(out_idx), N.ptr(seg_offset), st), "pbn_rulebook_pair_fill")
    hit = (in_idx, out_idx, seg_offset, n_seg, segment, seg_begin_d, totals)
    try:
        nbr._pbn_pairs = hit
    except AttributeError:
        pass
    return hit[:4]


def rulebook_pairs_dev(nbr, segment=WGRAD_PAIR_SEGMENT):
    """rulebook_pairs without the read-back: the lists are sized for the worst case (rows x offsets / segment + offsets
    segments), the per-offset segment ranges are computed on the device (pbn_rulebook_pair_fill_dev) and only consumers
    that read `seg_begin` on the device (pbn_spconv_wgrad) may use them.  -> (in_idx, out_idx, seg_begin, pair counts per
    offset (device))."""
    hit = getattr(nbr, "_pbn_pairs_dev", None)
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
    cap = (v * k) // segment + k
    in_idx = torch.empty(cap * segment, dtype=torch.int32, device=dev)
    out_idx = torch.empty(cap * segment, dtype=torch.int32, device=dev)
    seg_offset = torch.empty(cap, dtype=torch.int64, device=dev)
    seg_begin = torch.empty(k + 1, dtype=torch.int32, device=dev)
    N.check(lib.pbn_rulebook_pair_fill_dev(N.ptr(nbr), v, k, N.ptr(table), N.ptr(totals), segment, N.ptr(seg_begin),
                                           N.ptr(in_idx), N.ptr(out_idx), N.ptr(seg_offset), st), "pbn_rulebook_pair_fill_dev")
    hit = (in_idx, out_idx, seg_begin, totals, segment)
    try:
        nbr._pbn_pairs_dev = hit
    except AttributeError:
        pass
    return hit[:4]


def rulebook_pairs_dev_multi(maps, segment=WGRAD_PAIR_SEGMENT):
    """rulebook_pairs_dev of several maps in three launches (pbn_rulebook_pairs_multi; maps whose lists exist are skipped);
    the results are cached on the map tensors exactly as rulebook_pairs_dev caches them."""
    todo = [m for m in maps if getattr(m, "_pbn_pairs_dev", None) is None or m._pbn_pairs_dev[4] != segment]
    lib = N.lib()
    for i in range(0, len(todo), 16):
        group = todo[i:i + 16]
        jobs = (N.PairJob * len(group))()
        keep = []
        for j, nbr in enumerate(group):
            N.require_cuda(nbr)
            assert nbr.dtype == torch.int32 and nbr.is_contiguous()
            v, k = int(nbr.shape[0]), int(nbr.shape[1])
            dev = nbr.device
            table = torch.empty(max(lib.pbn_rulebook_pair_blocks(v), 1) * k, dtype=torch.int32, device=dev)
            totals = torch.empty(k, dtype=torch.int32, device=dev)
            cap = (v * k) // segment + k
            in_idx = torch.empty(cap * segment, dtype=torch.int32, device=dev)
            out_idx = torch.empty(cap * segment, dtype=torch.int32, device=dev)
            seg_offset = torch.empty(cap, dtype=torch.int64, device=dev)
            seg_begin = torch.empty(k + 1, dtype=torch.int32, device=dev)
            q = jobs[j]
            q.nbr, q.n, q.n_offsets = nbr.data_ptr(), v, k
            q.table, q.totals, q.seg_begin = table.data_ptr(), totals.data_ptr(), seg_begin.data_ptr()
            q.in_idx, q.out_idx, q.seg_offset = in_idx.data_ptr(), out_idx.data_ptr(), seg_offset.data_ptr()
            keep.append((nbr, (in_idx, out_idx, seg_begin, totals, segment), table, seg_offset))
        N.check(lib.pbn_rulebook_pairs_multi(jobs, len(group), segment, N.current_stream()), "pbn_rulebook_pairs_multi")
        for nbr, hit, _, _ in keep:
            nbr._pbn_pairs_dev = hit
    return [m._pbn_pairs_dev[:4] for m in maps]


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
    if int(feats.shape[0]) == 0 or int(grad_out.shape[0]) == 0:     # an empty level: no pairs, the gradient is zero
        return torch.zeros(1 if nbr is None else int(nbr.shape[1]), cin, cout, dtype=torch.float32, device=dev)
    if nbr is None:
        k, in_idx, out_idx, seg_begin, counts, n_pairs = 1, None, None, None, None, int(feats.shape[0])
    else:
        assert nbr.is_contiguous()
        k = int(nbr.shape[1])
        in_idx, out_idx, seg_begin, counts = rulebook_pairs_dev(nbr)   # no read-back: segment ranges and pair counts stay on the device
        # the host only needs the pair count to choose the number of pair splits (too few splits = too few workgroups, too
        # many = more partial slabs): half the table populated for cubes (the bench scene's stride-2..16 levels hold 12-16 of
        # 27, stride 1 7.6 -- where the split count is bounded by the workgroup target anyway), a quarter for the k=2 maps
        n_pairs = max(WGRAD_PAIR_SEGMENT, (int(nbr.shape[0]) * k) // (2 if k >= 27 else 4))
    dw = torch.empty(k, cin, cout, dtype=torch.float32, device=dev)
    ws = _WGRAD_WS.get(dev, int(lib.pbn_spconv_wgrad_workspace_bytes(k, cin, cout)))
    rc = lib.pbn_spconv_wgrad_checked(N.c_vp(feats.data_ptr()), feats.stride(0), int(feats.shape[0]), N.c_vp(grad_out.data_ptr()),
                                      grad_out.stride(0), int(grad_out.shape[0]), _DT[feats.dtype], N.ptr(in_idx), N.ptr(out_idx),
                                      N.ptr(seg_begin), N.ptr(counts), 0, WGRAD_PAIR_SEGMENT if nbr is not None else 0, n_pairs, k,
                                      int(cin), int(cout), N.ptr(dw), N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream())
    N.check(rc, "pbn_spconv_wgrad_checked")
    return dw


def _wgrad(feats, grad_out, nbr, cin, cout):
    return wgrad_native(feats, grad_out, nbr, cin, cout)


class _ConvFn(torch.autograd.Function):
    """Sparse convolution with autograd.  forward / dgrad run on the implicit-GEMM kernel (the input gradient of an
    output-stationary table is the same kind of table: the mirrored offset of the same map for odd kernels, the up table
    for a k=2,s=2 convolution, the down table for its transpose); wgrad is the pair-list contraction of csrc/wgrad.hip
    (wgrad_native)."""

    @staticmethod
    def forward(ctx, feats, kernel, bias, nbr, n_out, cache, dgrad_nbr, flip):
        packed = cache.get(kernel, feats.dtype)
        shift = None if bias is None else _pad_vec(bias, packed[3], 0.0)
        out = spconv_forward(feats, nbr, n_out, packed, shift=shift)
        cout = kernel.shape[-1]
        ctx.save_for_backward(feats, kernel)
        ctx.nbr, ctx.dgrad_nbr, ctx.flip, ctx.has_bias, ctx.cache = nbr, dgrad_nbr, flip, bias is not None, cache
        return out if out.shape[1] == cout else out[:, :cout]

    @staticmethod
    def backward(ctx, grad_out):
        feats, kernel = ctx.saved_tensors
        k3 = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
        cin, cout = k3.shape[1], k3.shape[2]
        grad_out = grad_out.contiguous()
        grad_feats = grad_kernel = grad_bias = None
        if ctx.needs_input_grad[0]:
            packed = ctx.cache.get_dgrad(kernel, grad_out.dtype, ctx.flip)                           # [K, Cout, Cin]
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
        self._cache = _PackCache(self)
        # the offsets of the input-gradient weights are mirrored for centred cubes only (see _map)
        self._dgrad_flip = (not is_transpose) and self.stride == 1 and self.kernel_size > 1 and self.kernel_size % 2 == 1
        _PACK_MODULES.add(self)
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
