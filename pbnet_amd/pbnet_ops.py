"""Drop-in for /root/reference/lib/PB_lib/torch_io/pbnet_ops.py (cluster / get_iou / cal_iou_and_masklabel /
get_normal_line), backed by libpbnet_hip.so.

Two layers:
  * the reference-shaped ops ``cluster``, ``get_iou``, ``cal_iou_and_masklabel`` (same names, positional
    arguments, return tuples and non-differentiability as pbnet_ops.py:82,111,141);
  * ``cluster_device`` -- the MI355X-native form the rebuilt PBNet.forward uses: device tensors in, device
    tensors out, any number of (class, batch) segments in ONE launch sequence, no host synchronisation.
"""
import torch
from torch.autograd import Function

from . import _native as N

PARA_F = 0.05     # pbnet_ops.py:70
NV_FLAG = True    # pbnet_ops.py:71


class ClusterResult(object):
    """Device-resident result of one grouping launch (capacity-sized; valid prefix given by n_clusters)."""
    __slots__ = ("cluster_id", "cluster_num", "den", "centers", "clt_sem", "n_clusters", "member_start",
                 "member_idx")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def cluster_device(off_xyz, org_xyz, sem, seg_len, radius, min_pts, para_f=PARA_F, nv_flag=NV_FLAG,
                   general_sem=False, want_members=True, capacity=False):
    """Grouping on device tensors.  off_xyz/org_xyz f32[I,3], sem i32[I], seg_len i32[B] (all CUDA).

    capacity=True: the row count of the inputs is only a capacity; the points that exist are the first sum(seg_len)
    rows (a device-side count) -- outputs beyond them are left untouched.

    Returns a ClusterResult of CUDA tensors; nothing is copied to the host and the stream is not synchronised.
    ``den`` is the neighbour count excluding self (binary.cu:148); ``cluster`` below adds the +1 of pbnet_ops.py:75.
    """
    N.require_cuda(off_xyz, org_xyz, sem, seg_len)
    off_xyz = off_xyz.to(torch.float32).contiguous()
    org_xyz = org_xyz.to(torch.float32).contiguous()
    sem = sem.to(torch.int32).contiguous()
    seg_len = seg_len.to(torch.int32).contiguous()
    n = int(off_xyz.shape[0])
    b = int(seg_len.shape[0])
    dev = off_xyz.device
    i32 = dict(dtype=torch.int32, device=dev)
    cluster_id = torch.empty(n, **i32)
    cluster_num = torch.empty(b, **i32)
    den = torch.empty(n, **i32)
    centers = torch.empty(3 * max(n, 1), dtype=torch.float32, device=dev)
    clt_sem = torch.empty(max(n, 1), **i32)
    n_clusters = torch.empty(1, **i32)
    member_start = torch.empty(n + 1, **i32) if want_members else None
    member_idx = torch.empty(max(n, 1), **i32) if want_members else None
    lib = N.lib()
    ws_bytes = lib.pbn_cluster_workspace_bytes(n, b, int(bool(general_sem)))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    rc = lib.pbn_binary_cluster(N.ptr(off_xyz), N.ptr(org_xyz), N.ptr(sem), N.ptr(seg_len), n, b, float(radius),
                                int(min_pts), float(para_f), int(bool(nv_flag)), int(bool(general_sem)) | (2 if capacity else 0),
                                N.ptr(cluster_id), N.ptr(cluster_num), N.ptr(den), N.ptr(centers), N.ptr(clt_sem),
                                N.ptr(n_clusters), N.ptr(member_start), N.ptr(member_idx), N.ptr(ws), ws_bytes,
                                N.current_stream())
    N.check(rc, "pbn_binary_cluster")
    return ClusterResult(cluster_id=cluster_id, cluster_num=cluster_num, den=den, centers=centers, clt_sem=clt_sem,
                         n_clusters=n_clusters, member_start=member_start, member_idx=member_idx)


class Cluster(Function):
    """pbnet_ops.py:12-79.  Accepts CPU tensors like the reference (they are staged to the current GPU) or CUDA
    tensors (zero-copy); returns tensors on the device of ``ins_offseted``."""

    @staticmethod
    def forward(ctx, ins_offseted, ins_orig, sem, ins_bp, radius, min_pts, batch_size):
        out_dev = ins_offseted.device
        dev = out_dev if out_dev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        sem_d = sem.to(dev)
        uniform = True
        if sem_d.numel() > 0:
            # one class per call is how PBNet.forward uses the op (PBNet.py:154,176); anything else takes the
            # general "component x class" path
            uniform = bool((sem_d == sem_d[0]).all().item())
        res = cluster_device(ins_offseted.to(dev), ins_orig.to(dev), sem_d, ins_bp.to(dev)[:batch_size], radius,
                             min_pts, general_sem=not uniform, want_members=False)
        c = int(res.n_clusters.item())
        if c < 0:
            raise RuntimeError("pbnet_ops.cluster: invalid input (segment lengths do not sum to the number of points, "
                               "or a class id outside [2,19])")
        center = res.centers[:3 * c]
        return (res.cluster_id.to(out_dev), res.cluster_num.to(out_dev), (res.den + 1).to(out_dev),
                center.to(out_dev))

    @staticmethod
    def backward(ctx, a=None, b=None, c=None, d=None):
        return None, None, None, None, None, None, None


cluster = Cluster.apply


def get_iou_device(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    N.require_cuda(proposals_idx, proposals_offset, instance_labels, instance_pointnum)
    n_inst = int(instance_pointnum.size(0))
    n_prop = int(proposals_offset.size(0)) - 1
    idx = proposals_idx.to(torch.int32).contiguous()
    off = proposals_offset.to(torch.int32).contiguous()
    lab = instance_labels.to(torch.int64).contiguous()
    pnum = instance_pointnum.to(torch.int32).contiguous()
    iou = torch.zeros(max(n_prop, 0), n_inst, dtype=torch.float32, device=idx.device)
    rc = N.lib().pbn_get_iou(N.ptr(idx), N.ptr(off), N.ptr(lab), N.ptr(pnum), N.ptr(iou), n_inst, n_prop,
                             N.current_stream())
    N.check(rc, "pbn_get_iou")
    return iou


class GetIoU(Function):
    """pbnet_ops.py:85-108."""

    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_pointnum):
        assert proposals_idx.is_contiguous() and proposals_idx.is_cuda
        assert proposals_offset.is_contiguous() and proposals_offset.is_cuda
        assert instance_labels.is_contiguous() and instance_labels.is_cuda
        assert instance_pointnum.is_contiguous() and instance_pointnum.is_cuda
        return get_iou_device(proposals_idx, proposals_offset, instance_labels, instance_pointnum)

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


get_iou = GetIoU.apply


class CalIoUAndMasklabel(Function):
    """pbnet_ops.py:114-138 (exported by the reference, never called by it)."""

    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_pointnum, mask_scores_sigmoid, mode):
        N.require_cuda(proposals_idx, proposals_offset, instance_labels, instance_pointnum, mask_scores_sigmoid)
        n_inst = int(instance_pointnum.size(0))
        n_prop = int(proposals_offset.size(0)) - 1
        idx = proposals_idx.to(torch.int32).contiguous()
        off = proposals_offset.to(torch.int32).contiguous()
        lab = instance_labels.to(torch.int64).contiguous()
        pnum = instance_pointnum.to(torch.int32).contiguous()
        ms = mask_scores_sigmoid.to(torch.float32).contiguous()
        iou = torch.zeros(n_prop, n_inst, dtype=torch.float32, device=idx.device)
        mask_label = torch.full(ms.shape, -1.0, dtype=torch.float32, device=idx.device)
        rc = N.lib().pbn_cal_iou_and_masklabel(N.ptr(idx), N.ptr(off), N.ptr(lab), N.ptr(pnum), N.ptr(iou), n_inst,
                                               n_prop, N.ptr(ms), N.ptr(mask_label), int(mode), N.current_stream())
        N.check(rc, "pbn_cal_iou_and_masklabel")
        return iou, mask_label

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


cal_iou_and_masklabel = CalIoUAndMasklabel.apply


def get_normal_line(xyz, face):
    """pbnet_ops.py:144-172.  Offline mesh preprocessing that the reference itself no longer calls
    (datasets/scannetv2/decode_scannet.py:113-117 is commented out); outside the hot path, not rebuilt."""
    raise NotImplementedError("cal_normal_line is offline preprocessing outside the MI355X hot path (SURVEY.md 2 #7)")
