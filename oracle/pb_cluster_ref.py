"""ctypes binding of oracle/pb_cluster_ref.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It mirrors the
reference call chain ``pbnet_ops.Cluster.forward`` -> ``PB_lib.binary_cluster``
(/root/reference/lib/PB_lib/torch_io/pbnet_ops.py:14-75, lib/PB_lib/src/pbnet/cluster.cu:16-119) on numpy arrays.
Parity unpinned: see the header of pb_cluster_ref.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libpbref.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpbref.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        lib.pbref_binary_cluster.restype = ctypes.c_int
        lib.pbref_binary_cluster.argtypes = [fp, fp, fp, fp, fp, fp, ip, ip, fp, ip, ip, ip, ip, fp, ip,
                                             ctypes.c_int, ctypes.c_float, ctypes.c_int, ip]
        lib.pbref_get_iou.restype = None
        lib.pbref_get_iou.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, ctypes.POINTER(ctypes.c_longlong), ip, fp]
        _LIB = lib
    return _LIB


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def binary_cluster(off_xyz, org_xyz, sem, seg_len, radius, min_pts, para_f=0.05, nv_flag=True):
    """Raw oracle call.  Returns dict(cluster_id, cluster_num, den_queue, center[C,3], clt_sem[C]).

    ``den_queue`` is the neighbour count EXCLUDING self, as written by binary.cu:148; the Python wrapper of the
    reference adds 1 (pbnet_ops.py:75) -- see :func:`cluster`.
    """
    off_xyz = np.ascontiguousarray(off_xyz, dtype=np.float32).reshape(-1, 3)
    org_xyz = np.ascontiguousarray(org_xyz, dtype=np.float32).reshape(-1, 3)
    n = off_xyz.shape[0]
    x, y, z = (np.ascontiguousarray(off_xyz[:, k]) for k in range(3))
    xo, yo, zo = (np.ascontiguousarray(org_xyz[:, k]) for k in range(3))
    sem = np.ascontiguousarray(sem, dtype=np.int32)
    seg_len = np.ascontiguousarray(seg_len, dtype=np.int32)
    assert int(seg_len.sum()) == n and sem.shape[0] == n
    radius18 = np.full(18, radius, dtype=np.float32)      # pbnet_ops.py:33-34
    min_pts18 = np.full(18, min_pts, dtype=np.int32)      # pbnet_ops.py:35-36
    cluster_id = np.full(n, -1, dtype=np.int32)           # pbnet_ops.py:40-41
    cluster_num = np.zeros(seg_len.shape[0], dtype=np.int32)
    den = np.zeros(n, dtype=np.int32)
    center = np.zeros(3 * max(n, 1), dtype=np.float32)
    clt_sem = np.zeros(max(n, 1), dtype=np.int32)
    n_clusters = ctypes.c_int(0)
    rc = _lib().pbref_binary_cluster(_f(x), _f(y), _f(z), _f(xo), _f(yo), _f(zo), _i(sem), _i(seg_len),
                                     _f(radius18), _i(min_pts18), _i(cluster_id), _i(cluster_num), _i(den),
                                     _f(center), _i(clt_sem), int(seg_len.shape[0]), float(np.float32(para_f)),
                                     int(bool(nv_flag)), ctypes.byref(n_clusters))
    if rc != 0:
        raise RuntimeError("pbref_binary_cluster failed with code %d" % rc)
    c = n_clusters.value
    return dict(cluster_id=cluster_id, cluster_num=cluster_num, den_queue=den,
                center=center[:3 * c].reshape(c, 3).copy(), clt_sem=clt_sem[:c].copy())


def cluster(ins_offseted, ins_orig, sem, ins_bp, radius, min_pts, batch_size=None):
    """``pbnet_ops.cluster`` semantics (pbnet_ops.py:14-75): returns (cluster_id, cluster_num, den+1, center flat)."""
    out = binary_cluster(ins_offseted, ins_orig, sem, ins_bp, radius, min_pts, para_f=0.05, nv_flag=True)
    return out["cluster_id"], out["cluster_num"], out["den_queue"] + 1, out["center"].reshape(-1)


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    """``pbnet_ops.get_iou`` semantics (pbnet_ops.py:85-111, get_iou.cu:12-38)."""
    proposals_idx = np.ascontiguousarray(proposals_idx, dtype=np.int32)
    proposals_offset = np.ascontiguousarray(proposals_offset, dtype=np.int32)
    instance_labels = np.ascontiguousarray(instance_labels, dtype=np.int64)
    instance_pointnum = np.ascontiguousarray(instance_pointnum, dtype=np.int32)
    n_inst = instance_pointnum.shape[0]
    n_prop = proposals_offset.shape[0] - 1
    out = np.zeros((n_prop, n_inst), dtype=np.float32)
    _lib().pbref_get_iou(n_inst, n_prop, _i(proposals_idx), _i(proposals_offset),
                         instance_labels.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                         _i(instance_pointnum), _f(out))
    return out
