"""Device coordinate manager and SparseTensor: the subset of MinkowskiEngine's tensor API that the reference
touches (/root/reference/network/PBNet.py:117,125-128,240-250,265-271; network/Mink.py:291-354), rebuilt on
libpbnet_hip.so (csrc/coords.hip).  MinkowskiEngine itself is an un-vendored, un-pinned third-party dependency of
the reference (README.md:15-27); the behavioural conventions assumed here are listed in conventions.py.
"""
import ctypes

import torch

from .. import _native as N
from . import conventions as CV


def _i32(n, dev, fill=None):
    if fill is None:
        return torch.empty(n, dtype=torch.int32, device=dev)
    return torch.full((n,), fill, dtype=torch.int32, device=dev)


_OFFSETS = {}
_RANGE_ERROR = ("coordinates cannot be indexed: batch must be in [0,65534], x/y/z in [-32768,32767], and the bounding box of one "
                "call must fit 44 Z-order key bits (3 x bits of the largest extent + bits of the batch range); also raised when a "
                "coordinate table was found full or corrupted")


def _device_offsets(kernel_size, stride, device):
    """Kernel offsets live on the device once per (kernel, stride): no per-forward host->device copy."""
    key = (int(kernel_size), int(stride), device)
    t = _OFFSETS.get(key)
    if t is None:
        t = CV.kernel_offsets(kernel_size, stride).to(device)
        _OFFSETS[key] = t
    return t


class _Level(object):
    """One tensor stride of a coordinate pyramid: coordinates, hash table, parent links to the next coarser level."""
    __slots__ = ("stride", "coords", "n", "n_dev", "keys", "vals", "capacity", "parent_row", "child_k", "nbr_down")


class _Pyramid(object):
    """One coordinate order of a lineage, fully expanded by ONE native call (pbn_coords_build): the four coarser levels,
    the k=3 maps of all levels, the k=5 map of level 1 and the transposed-convolution tables, in one arena with
    device-resident row counts."""
    _STRIDES = (1, 2, 4, 8, 16)

    def __init__(self, coords, n_dev, n_cap, _prepared=None):
        lib = N.lib()
        if _prepared is not None:           # (arena, layout) already filled by pbn_coords_prepare
            self.arena, self.layout = _prepared
            self.device = self.arena.device
        else:
            self.device = coords.device
            self.layout = N.CoordsLayout()
            nbytes = lib.pbn_coords_arena_bytes(n_cap, 1, ctypes.byref(self.layout))
            self.arena = torch.empty(nbytes, dtype=torch.uint8, device=coords.device)
            rc = lib.pbn_coords_build(N.ptr(coords), None if n_dev is None else N.ptr(n_dev), n_cap, 1,
                                      int(CV.X_FASTEST), N.ptr(self.arena), nbytes, ctypes.byref(self.layout),
                                      N.current_stream())
            N.check(rc, "pbn_coords_build")
        self.counts_dev = self.view(self.layout.counts, 5, torch.int32)
        self.n = None

    def view(self, offset, count, dtype, shape=None):
        nbytes = count * torch.empty(0, dtype=dtype).element_size()
        t = self.arena[offset:offset + nbytes].view(dtype)
        return t if shape is None else t.view(shape)

    def ptr(self, offset):
        return self.arena.data_ptr() + offset

    def set_counts(self, counts):
        if min(counts) < 0:
            raise ValueError(_RANGE_ERROR)
        self.n = [int(c) for c in counts]

    def finalize(self):
        if self.n is None:
            self.set_counts(self.counts_dev.tolist())

    def level_index(self, stride):
        return self._STRIDES.index(stride)

    def coordinates(self, stride):
        self.finalize()
        l = self.level_index(stride)
        return self.view(self.layout.coords[l], self.n[l] * 4, torch.int32, (self.n[l], 4))

    def _map_view(self, key, make):
        """One tensor object per map for the life of the pyramid: per-map derived tables (the weight gradient's pair
        lists) are cached on the object and shared by every layer of the level."""
        views = self.__dict__.setdefault("_views", {})
        if key not in views:
            views[key] = make()
        return views[key]

    def kernel_map(self, stride, kernel_size):
        self.finalize()
        l = self.level_index(stride)
        if kernel_size == 3:
            return self._map_view(("k3", l), lambda: self.view(self.layout.k3[l], self.n[l] * 27, torch.int32, (self.n[l], 27)))
        assert kernel_size == 5 and stride == 1
        return self._map_view(("k5",), lambda: self.view(self.layout.k5, self.n[0] * 125, torch.int32, (self.n[0], 125)))

    def down_map(self, stride_in):
        self.finalize()
        l = self.level_index(stride_in)
        return self._map_view(("down", l), lambda: self.view(self.layout.nbr_down[l], self.n[l + 1] * 8, torch.int32,
                                                             (self.n[l + 1], 8)))

    def up_map(self, stride_in):
        self.finalize()
        l = self.level_index(stride_in) - 1
        return self._map_view(("up", l), lambda: self.view(self.layout.up[l], self.n[l] * 8, torch.int32, (self.n[l], 8)))

    def native_tables(self):
        self.finalize()
        L = self.layout
        return ([self.ptr(L.k3[l]) for l in range(5)], self.ptr(L.k5), [self.ptr(L.nbr_down[l]) for l in range(4)],
                [self.ptr(L.up[l]) for l in range(4)])


class SortedView(object):
    """The lineage in Morton (Z) order: `perm[p]` = external row at sorted position p, `inv_perm` its inverse."""
    __slots__ = ("pyramid", "perm", "inv_perm")


class CoordinateManager(object):
    """Owns the coordinate sets of one SparseTensor lineage and the kernel maps between them (ME caches both per
    lineage in its coordinate manager, so transposed convolutions land exactly on the encoder's coordinates).

    External row order = survivors of the de-duplication in ascending original order (ME semantics).  Two expansions
    of the lineage exist, each built by one native call and only when asked for:
      * `plain`  -- maps in the external row order (module-by-module path, training);
      * `sorted` -- the same lineage in Morton order (fused inference path): a 128-row tile is then a compact spatial
                    block, so whole kernel offsets drop out of a tile and its gathers stay L2-local.  Coarser levels
                    inherit the order (parents are numbered by first occurrence).
    `prepare` selects what is launched up front so that all row counts come back in ONE host read."""

    MAX_STRIDE = 16
    _STRIDES = (1, 2, 4, 8, 16)

    def __init__(self, coordinates, build_maps=True, prepare=None):
        N.require_cuda(coordinates)
        coords = coordinates.to(torch.int32).contiguous()
        assert coords.dim() == 2 and coords.shape[1] == 4, "coordinates must be [N,4] (batch, x, y, z)"
        dev = coords.device
        lib = N.lib()
        n = int(coords.shape[0])
        self.device = dev
        self.n_input = n
        self._maps = {}
        self._plain = None
        self._sorted = None
        self._final = False
        self.unique_index = self.inverse_mapping = self.is_identity = None
        self._n1 = None
        if prepare is None:
            prepare = ("plain" if torch.is_grad_enabled() else "sorted") if build_maps else "unique"
        self._native = None
        if prepare == "sorted" and n > 0:
            # inference: de-duplication + Z-order + pyramid + maps in ONE native call (pbn_coords_prepare)
            P = N.PrepareLayout()
            nbytes = lib.pbn_coords_prepare_bytes(n, 1, ctypes.byref(P))
            arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.pbn_coords_prepare(N.ptr(coords), n, 1, int(CV.X_FASTEST), N.ptr(arena), nbytes, ctypes.byref(P),
                                        N.current_stream())
            N.check(rc, "pbn_coords_prepare")
            self._native = (arena, P)
            self._prepare_input = coords            # kept until the counts are read: a key-bits failure retries through the hash pipeline
            sv = SortedView()
            sv.pyramid = _Pyramid(None, None, n, _prepared=(arena, P.pyramid))
            sv.perm = sv.inv_perm = None
            self._sorted = sv
            self._count = sv.pyramid.view(P.n_unique, 1, torch.int32)
            self._ucoords = sv.pyramid.view(P.ucoords, max(n, 1) * 4, torch.int32, (max(n, 1), 4))
            return
        cap = lib.pbn_hash_capacity(n)
        keys = torch.empty(cap, dtype=torch.int64, device=dev)
        vals = _i32(cap, dev)
        self._count = _i32(1, dev)
        self._uidx, self._inv = _i32(max(n, 1), dev), _i32(max(n, 1), dev)
        self._ucoords = torch.empty(max(n, 1), 4, dtype=torch.int32, device=dev)
        wsb = lib.pbn_coords_workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.pbn_coords_unique(N.ptr(coords), None, n, N.ptr(keys), N.ptr(vals), cap, N.ptr(self._uidx),
                                   N.ptr(self._inv), N.ptr(self._ucoords), N.ptr(self._count), N.ptr(ws), wsb,
                                   N.current_stream())
        N.check(rc, "pbn_coords_unique")
        if prepare == "plain":
            self._build_plain()
        elif prepare == "sorted":
            self._build_sorted()

    # -- expansions ---------------------------------------------------------------------------------------------
    def _build_plain(self):
        if self._plain is None:
            self._plain = _Pyramid(self._ucoords, self._count, self.n_input)
        return self._plain

    def _build_sorted(self):
        if self._sorted is None:
            n = self.n_input
            keys = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
            rc = N.lib().pbn_morton_keys(N.ptr(self._ucoords), N.ptr(self._count), n, N.ptr(keys), N.current_stream())
            N.check(rc, "pbn_morton_keys")
            perm = torch.sort(keys[:n], stable=True)[1]            # padding rows (beyond the unique count) sort last
            sv = SortedView()
            sv.perm = perm
            sv.pyramid = _Pyramid(self._ucoords[perm].contiguous(), self._count, n)
            sv.inv_perm = None
            self._sorted = sv
        return self._sorted

    def _finalize(self):
        if self._final:
            return
        if self._native is not None:
            arena, P = self._native
            pyr = self._sorted.pyramid
            counts = pyr.counts_dev.tolist()                                           # the one host read
            if min(counts) < 0 and self.__dict__.get("_prepare_input") is not None:
                # the sorted pipeline refuses boxes whose Z-order key needs more than 44 bits (very wide scenes, large batch
                # ranges); the hash-table pipeline takes any in-range 16-bit coordinates and fills the same arena and layout
                coords = self._prepare_input
                rc = N.lib().pbn_coords_prepare_hash(N.ptr(coords), None, self.n_input, 1, int(CV.X_FASTEST), N.ptr(arena),
                                                     arena.numel(), ctypes.byref(P), N.current_stream())
                N.check(rc, "pbn_coords_prepare_hash")
                counts = pyr.counts_dev.tolist()
            self._prepare_input = None
            pyr.set_counts(counts)
            n1 = self._n1 = int(counts[0])
            n = self.n_input
            self.unique_index = pyr.view(P.unique_index, n, torch.int64)[:n1]
            self.inverse_mapping = pyr.view(P.inverse, n, torch.int64)
            self.is_identity = (n1 == n)
            self._sorted.perm = pyr.view(P.perm, n, torch.int64)[:n1]
            self._sorted.inv_perm = pyr.view(P.inv_perm, n, torch.int64)[:n1]
            self._final = True
            return
        pending = [p for p in (self._plain, self._sorted.pyramid if self._sorted else None) if p is not None and p.n is None]
        counts = torch.cat([self._count] + [p.counts_dev for p in pending]).tolist()   # the one host read
        if min(counts) < 0:
            raise ValueError(_RANGE_ERROR)
        self._n1 = int(counts[0])
        for i, p in enumerate(pending):
            p.set_counts(counts[1 + 5 * i:6 + 5 * i])
        n1 = self._n1
        self.unique_index = self._uidx[:n1].long()
        self.inverse_mapping = self._inv[:self.n_input].long()
        self.is_identity = (n1 == self.n_input)
        if self._sorted is not None:
            sv = self._sorted
            sv.perm = sv.perm[:n1]
            sv.inv_perm = torch.empty_like(sv.perm)
            sv.inv_perm[sv.perm] = torch.arange(n1, device=self.device)
        self._final = True

    def plain(self):
        self._finalize()
        if self._plain is None:
            self._build_plain()
        self._plain.finalize()
        return self._plain

    def sorted(self):
        self._finalize()
        if self._sorted is None:
            self._build_sorted()
            sv = self._sorted
            sv.pyramid.finalize()
            sv.perm = sv.perm[:self._n1]
            sv.inv_perm = torch.empty_like(sv.perm)
            sv.inv_perm[sv.perm] = torch.arange(self._n1, device=self.device)
        return self._sorted

    # -- sizes ------------------------------------------------------------------------------------------------
    def num_rows(self, stride):
        self._finalize()
        if stride == 1:
            return self._n1
        pyr = self._plain if self._plain is not None else (self._sorted.pyramid if self._sorted else self.plain())
        pyr.finalize()
        return pyr.n[self._STRIDES.index(stride)]

    def row_counts(self):
        return [self.num_rows(s) for s in self._STRIDES]

    def coordinates(self, stride):
        self._finalize()
        if stride == 1:
            return self._ucoords[:self._n1]
        return self.plain().coordinates(stride)

    # -- kernel maps in the external row order --------------------------------------------------------------------
    def kernel_map(self, stride, kernel_size):
        """nbr[n(stride), K^3] for a stride-1 (in the tensor-stride sense) convolution of odd kernel size."""
        pyr = self.plain()
        if kernel_size == 3 or (kernel_size == 5 and stride == 1):
            return pyr.kernel_map(stride, kernel_size)
        key = ("k", stride, kernel_size)
        if key not in self._maps:
            l = pyr.level_index(stride)
            n = pyr.n[l]
            L = pyr.layout
            nbr = torch.empty(max(n, 1), kernel_size ** 3, dtype=torch.int32, device=self.device)
            rc = N.lib().pbn_kernel_map_cube(N.c_vp(pyr.ptr(L.coords[l])), None, n, kernel_size, stride,
                                             int(CV.X_FASTEST), N.c_vp(pyr.ptr(L.keys[l])), N.c_vp(pyr.ptr(L.vals[l])),
                                             L.capacity[l], N.ptr(nbr), N.current_stream())
            N.check(rc, "pbn_kernel_map_cube")
            self._maps[key] = nbr[:n]
        return self._maps[key]

    def down_map(self, stride_in):
        """k=2,s=2 convolution stride_in -> 2*stride_in: nbr_down[n_coarse, 8] child rows."""
        return self.plain().down_map(stride_in)

    def up_map(self, stride_in):
        """Transposed k=2,s=2 convolution stride_in -> stride_in/2: nbr_up[n_fine, 8]."""
        return self.plain().up_map(stride_in)


class SparseTensor(object):
    """ME.SparseTensor subset: .F, .C, .inverse_mapping, .coordinate_manager, .tensor_stride.

    Construction from (features, coordinates) de-duplicates like ME's default RANDOM_SUBSAMPLE mode (one row per
    voxel survives -- here deterministically the first occurrence; unique input keeps its row order, which
    PBNet.forward relies on at PBNet.py:130)."""

    def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_manager=None, device=None,
                 quantization_mode=None, _slab=None):
        if device is not None:
            dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
            features = features.to(dev)
            if coordinates is not None:
                coordinates = coordinates.to(dev)
        if coordinate_manager is None:
            assert coordinates is not None
            N.require_cuda(features, coordinates)
            cm = CoordinateManager(coordinates)
            cm.num_rows(1)
            if not cm.is_identity:
                features = features[cm.unique_index]
            coordinate_manager = cm
            tensor_stride = 1
        self._F = features
        self._stored = None
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = int(tensor_stride)

    @classmethod
    def _from_stored_rows(cls, stored, row_index, coordinate_manager, tensor_stride=1):
        """Features kept in another row order (the fused U-Net computes in Z-order): row i of this tensor is
        stored[row_index[i]].  `.F` materialises that gather on first use; consumers that index rows anyway take
        `rows()` and fold `row_index` into their own index instead."""
        self = cls.__new__(cls)
        self._F = None
        self._stored = (stored, row_index)
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = int(tensor_stride)
        return self

    def rows(self):
        """(features, row_index): row i is features[row_index[i]] (row_index None = rows are in place)."""
        if self._F is None:
            return self._stored
        return self._F, None

    @property
    def F(self):
        if self._F is None:
            stored, row_index = self._stored
            self._F = stored[row_index]
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.coordinates(self.tensor_stride)

    @property
    def inverse_mapping(self):
        return self.coordinate_manager.inverse_mapping

    @property
    def device(self):
        return self.rows()[0].device

    @property
    def dtype(self):
        return self.rows()[0].dtype

    @property
    def shape(self):
        f, idx = self.rows()
        return f.shape if idx is None else torch.Size((idx.shape[0], f.shape[1]))

    def replace_feature(self, feats):
        return SparseTensor(feats, coordinate_manager=self.coordinate_manager, tensor_stride=self.tensor_stride)

    def __repr__(self):
        return "SparseTensor(F=%s, stride=%d)" % (tuple(self.shape), self.tensor_stride)


def cat(*tensors):
    """ME.cat (Mink.py:323,331,339,347): feature concatenation of tensors sharing a coordinate map."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    t0 = tensors[0]
    for t in tensors[1:]:
        assert t.coordinate_manager is t0.coordinate_manager and t.tensor_stride == t0.tensor_stride
    return t0.replace_feature(torch.cat([t.F for t in tensors], dim=1))
