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


class CoordinateManager(object):
    """Owns the coordinate sets of one SparseTensor lineage and the kernel maps between them (ME caches both per
    lineage in its coordinate manager, so transposed convolutions land exactly on the encoder's coordinates).

    Construction is ONE native call (pbn_coords_build): de-duplication, the four coarser levels, the k=3 maps of all
    levels, the k=5 map of level 1 and the transposed-convolution tables, laid out in one arena with device-resident
    row counts.  The five counts come back in ONE host read the first time a size is needed."""

    MAX_STRIDE = 16
    _STRIDES = (1, 2, 4, 8, 16)

    def __init__(self, coordinates, build_maps=True):
        N.require_cuda(coordinates)
        coords = coordinates.to(torch.int32).contiguous()
        assert coords.dim() == 2 and coords.shape[1] == 4, "coordinates must be [N,4] (batch, x, y, z)"
        dev = coords.device
        lib = N.lib()
        n = int(coords.shape[0])
        self.device = dev
        self.n_input = n
        self.build_maps = build_maps
        self._maps = {}
        self._final = False
        self.unique_index = self.inverse_mapping = self.is_identity = None
        self._n = None
        if build_maps:
            self._layout = N.CoordsLayout()
            nbytes = lib.pbn_coords_arena_bytes(n, 1, ctypes.byref(self._layout))
            self._arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.pbn_coords_build(N.ptr(coords), n, 1, int(CV.X_FASTEST), N.ptr(self._arena), nbytes,
                                      ctypes.byref(self._layout), N.current_stream())
            N.check(rc, "pbn_coords_build")
            self._counts = self._view(self._layout.counts, 5, torch.int32)
        else:  # de-duplication only (ME.utils.sparse_quantize)
            cap = lib.pbn_hash_capacity(n)
            self._keys = torch.empty(cap, dtype=torch.int64, device=dev)
            self._vals = _i32(cap, dev)
            self._counts = _i32(1, dev)
            self._uidx, self._inv = _i32(max(n, 1), dev), _i32(max(n, 1), dev)
            self._ucoords = torch.empty(max(n, 1), 4, dtype=torch.int32, device=dev)
            wsb = lib.pbn_coords_workspace_bytes(n)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            rc = lib.pbn_coords_unique(N.ptr(coords), None, n, N.ptr(self._keys), N.ptr(self._vals), cap,
                                       N.ptr(self._uidx), N.ptr(self._inv), N.ptr(self._ucoords), N.ptr(self._counts),
                                       N.ptr(ws), wsb, N.current_stream())
            N.check(rc, "pbn_coords_unique")

    def _view(self, offset, count, dtype, shape=None):
        nbytes = count * torch.empty(0, dtype=dtype).element_size()
        t = self._arena[offset:offset + nbytes].view(dtype)
        return t if shape is None else t.view(shape)

    def _ptr(self, offset):
        return self._arena.data_ptr() + offset

    # -- sizes ------------------------------------------------------------------------------------------------
    def _finalize(self):
        if self._final:
            return
        counts = self._counts.tolist()  # the one host synchronisation of this lineage
        if counts[0] < 0:
            raise ValueError("coordinate out of range: batch must be in [0,65534], x/y/z in [-32768,32767]")
        self._n = [int(c) for c in counts]
        n1 = self._n[0]
        if self.build_maps:
            self.unique_index = self._view(self._layout.unique_index, n1, torch.int32).long()
            self.inverse_mapping = self._view(self._layout.inverse, self.n_input, torch.int32).long()
        else:
            self.unique_index = self._uidx[:n1].long()
            self.inverse_mapping = self._inv[:self.n_input].long()
        self.is_identity = (n1 == self.n_input)
        self._final = True

    def _level_index(self, stride):
        assert self.build_maps, "this coordinate manager was built for de-duplication only"
        return self._STRIDES.index(stride)

    def num_rows(self, stride):
        self._finalize()
        return self._n[self._STRIDES.index(stride)] if self.build_maps else self._n[0]

    def row_counts(self):
        self._finalize()
        return list(self._n)

    def coordinates(self, stride):
        self._finalize()
        if not self.build_maps:
            return self._ucoords[:self._n[0]]
        l = self._level_index(stride)
        return self._view(self._layout.coords[l], self._n[l] * 4, torch.int32, (self._n[l], 4))

    # -- kernel maps (views into the arena; rows beyond the level's count are never touched) ---------------------
    def kernel_map(self, stride, kernel_size):
        """nbr[n(stride), K^3] for a stride-1 (in the tensor-stride sense) convolution of odd kernel size."""
        self._finalize()
        l = self._level_index(stride)
        if kernel_size == 3:
            return self._view(self._layout.k3[l], self._n[l] * 27, torch.int32, (self._n[l], 27))
        if kernel_size == 5 and stride == 1:
            return self._view(self._layout.k5, self._n[0] * 125, torch.int32, (self._n[0], 125))
        key = ("k", stride, kernel_size)
        if key not in self._maps:
            n = self._n[l]
            nbr = torch.empty(max(n, 1), kernel_size ** 3, dtype=torch.int32, device=self.device)
            rc = N.lib().pbn_kernel_map_cube(N.c_vp(self._ptr(self._layout.coords[l])), None, n, kernel_size, stride,
                                             int(CV.X_FASTEST), N.c_vp(self._ptr(self._layout.keys[l])),
                                             N.c_vp(self._ptr(self._layout.vals[l])), self._layout.capacity[l],
                                             N.ptr(nbr), N.current_stream())
            N.check(rc, "pbn_kernel_map_cube")
            self._maps[key] = nbr[:n]
        return self._maps[key]

    def down_map(self, stride_in):
        """k=2,s=2 convolution stride_in -> 2*stride_in: nbr_down[n_coarse, 8] child rows."""
        self._finalize()
        l = self._level_index(stride_in)
        return self._view(self._layout.nbr_down[l], self._n[l + 1] * 8, torch.int32, (self._n[l + 1], 8))

    def up_map(self, stride_in):
        """Transposed k=2,s=2 convolution stride_in -> stride_in/2: nbr_up[n_fine, 8]."""
        self._finalize()
        l = self._level_index(stride_in) - 1
        return self._view(self._layout.up[l], self._n[l] * 8, torch.int32, (self._n[l], 8))

    def native_tables(self):
        """Raw device addresses for the native U-Net executor: (k3[5], k5, down[4], up[4])."""
        self._finalize()
        L = self._layout
        return ([self._ptr(L.k3[l]) for l in range(5)], self._ptr(L.k5), [self._ptr(L.nbr_down[l]) for l in range(4)],
                [self._ptr(L.up[l]) for l in range(4)])


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
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = int(tensor_stride)

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.coordinates(self.tensor_stride)

    @property
    def inverse_mapping(self):
        return self.coordinate_manager.inverse_mapping

    @property
    def device(self):
        return self._F.device

    @property
    def dtype(self):
        return self._F.dtype

    @property
    def shape(self):
        return self._F.shape

    def replace_feature(self, feats):
        return SparseTensor(feats, coordinate_manager=self.coordinate_manager, tensor_stride=self.tensor_stride)

    def __repr__(self):
        return "SparseTensor(F=%s, stride=%d)" % (tuple(self._F.shape), self.tensor_stride)


def cat(*tensors):
    """ME.cat (Mink.py:323,331,339,347): feature concatenation of tensors sharing a coordinate map."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    t0 = tensors[0]
    for t in tensors[1:]:
        assert t.coordinate_manager is t0.coordinate_manager and t.tensor_stride == t0.tensor_stride
    return t0.replace_feature(torch.cat([t.F for t in tensors], dim=1))
