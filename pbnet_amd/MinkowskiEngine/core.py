"""Device coordinate manager and SparseTensor: the subset of MinkowskiEngine's tensor API that the reference
touches (/root/reference/network/PBNet.py:117,125-128,240-250,265-271; network/Mink.py:291-354), rebuilt on
libpbnet_hip.so (csrc/coords.hip).  MinkowskiEngine itself is an un-vendored, un-pinned third-party dependency of
the reference (README.md:15-27); the behavioural conventions assumed here are listed in conventions.py.
"""
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

    Construction launches the de-duplication AND the four coarser levels back to back on device-resident row counts;
    the five counts come back in ONE host read the first time a size is needed."""

    MAX_STRIDE = 16

    def __init__(self, coordinates):
        N.require_cuda(coordinates)
        coords = coordinates.to(torch.int32).contiguous()
        assert coords.dim() == 2 and coords.shape[1] == 4, "coordinates must be [N,4] (batch, x, y, z)"
        dev = coords.device
        lib = N.lib()
        n = int(coords.shape[0])
        self.device = dev
        self.n_input = n
        self._maps = {}
        self._levels = {}
        self._counts = torch.empty(5, dtype=torch.int32, device=dev)
        ws_bytes = lib.pbn_coords_workspace_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        stream = N.current_stream()
        cap0 = max(n, 1)

        def new_level(stride, idx):
            lv = _Level()
            lv.stride = stride
            lv.capacity = lib.pbn_hash_capacity(n)
            lv.keys = torch.empty(lv.capacity, dtype=torch.int64, device=dev)
            lv.vals = _i32(lv.capacity, dev)
            lv.n_dev = self._counts[idx:idx + 1]
            lv.coords = torch.empty(cap0, 4, dtype=torch.int32, device=dev)
            lv.n = None
            lv.parent_row = lv.child_k = lv.nbr_down = None
            self._levels[stride] = lv
            return lv

        lv = new_level(1, 0)
        self._unique_index = _i32(cap0, dev)
        self._inverse = _i32(cap0, dev)
        rc = lib.pbn_coords_unique(N.ptr(coords), None, n, N.ptr(lv.keys), N.ptr(lv.vals), lv.capacity,
                                   N.ptr(self._unique_index), N.ptr(self._inverse), N.ptr(lv.coords), N.ptr(lv.n_dev),
                                   N.ptr(ws), ws_bytes, stream)
        N.check(rc, "pbn_coords_unique")
        fine, s, idx = lv, 2, 1
        while s <= self.MAX_STRIDE:
            lv = new_level(s, idx)
            fine.parent_row = _i32(cap0, dev)
            fine.child_k = _i32(cap0, dev)
            fine.nbr_down = torch.empty(cap0, 8, dtype=torch.int32, device=dev)
            rc = lib.pbn_coords_stride(N.ptr(fine.coords), N.ptr(fine.n_dev), n, s, N.ptr(lv.keys), N.ptr(lv.vals),
                                       lv.capacity, N.ptr(lv.coords), N.ptr(fine.parent_row), N.ptr(fine.child_k),
                                       N.ptr(fine.nbr_down), N.ptr(lv.n_dev), N.ptr(ws), ws_bytes, stream)
            N.check(rc, "pbn_coords_stride")
            fine, s, idx = lv, s * 2, idx + 1
        self._ws = ws
        self._final = False
        self.unique_index = None
        self.inverse_mapping = None
        self.is_identity = None

    # -- sizes ------------------------------------------------------------------------------------------------
    def _finalize(self):
        if self._final:
            return
        counts = self._counts.tolist()  # the one host synchronisation of this lineage
        if counts[0] < 0:
            raise ValueError("coordinate out of range: batch must be in [0,65534], x/y/z in [-32768,32767]")
        prev = None
        for lv, c in zip((self._levels[s] for s in (1, 2, 4, 8, 16)), counts):
            lv.n = int(c)
            lv.coords = lv.coords[:lv.n]
            if prev is not None:
                prev.parent_row = prev.parent_row[:prev.n]
                prev.child_k = prev.child_k[:prev.n]
                prev.nbr_down = prev.nbr_down[:lv.n]
            prev = lv
        n1 = self._levels[1].n
        self.unique_index = self._unique_index[:n1].long()
        self.inverse_mapping = self._inverse[:self.n_input].long()
        self.is_identity = (n1 == self.n_input)
        self._final = True

    def _build_pyramid(self):
        self._finalize()

    def level(self, stride):
        self._finalize()
        return self._levels[stride]

    def num_rows(self, stride):
        return self.level(stride).n

    def coordinates(self, stride):
        return self.level(stride).coords

    # -- kernel maps ------------------------------------------------------------------------------------------
    def kernel_map(self, stride, kernel_size):
        """nbr[n(stride), K^3] for a stride-1 (in the tensor-stride sense) convolution of odd kernel size."""
        key = ("k", stride, kernel_size)
        if key not in self._maps:
            lv = self.level(stride)
            off = _device_offsets(kernel_size, stride, self.device)
            k = int(off.shape[0])
            nbr = torch.empty(max(lv.n, 1), k, dtype=torch.int32, device=self.device)
            rc = N.lib().pbn_kernel_map(N.ptr(lv.coords), None, lv.n, N.ptr(off), k, N.ptr(lv.keys), N.ptr(lv.vals),
                                        lv.capacity, N.ptr(nbr), N.current_stream())
            N.check(rc, "pbn_kernel_map")
            self._maps[key] = nbr[:lv.n]
        return self._maps[key]

    def down_map(self, stride_in):
        """k=2,s=2 convolution stride_in -> 2*stride_in: nbr_down[n_coarse, 8] child rows."""
        self._build_pyramid()
        return self._levels[stride_in].nbr_down

    def up_map(self, stride_in):
        """Transposed k=2,s=2 convolution stride_in -> stride_in/2: nbr_up[n_fine, 8]."""
        key = ("u", stride_in)
        if key not in self._maps:
            self._build_pyramid()
            fine = self._levels[stride_in // 2]
            nbr = torch.empty(max(fine.n, 1), 8, dtype=torch.int32, device=self.device)
            rc = N.lib().pbn_up_table(N.ptr(fine.parent_row), N.ptr(fine.child_k), None, fine.n, N.ptr(nbr),
                                      N.current_stream())
            N.check(rc, "pbn_up_table")
            self._maps[key] = nbr[:fine.n]
        return self._maps[key]


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
            cm.level(1)
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
