"""Drop-in for the reference's native module ``PB_lib`` (/root/reference/lib/PB_lib/src/PB_lib_api.cpp:6-11):
the same four callables with the same positional signatures and in-place output conventions, served by
libpbnet_hip.so.  ``import pbnet_amd.PB_lib as PB_lib`` replaces ``import PB_lib``.

The reference entry points take CPU tensors for binary_cluster (cluster.h:13-18) and CUDA tensors for get_iou
(get_iou.h:15); both kinds are accepted here -- CPU tensors are staged through the current GPU.
"""
import torch

from . import _native as N
from . import pbnet_ops as _ops


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def binary_cluster(x, y, z, l1_norm, index_mapper, xo, yo, zo, sem, batch_index, radius, min_pts, cluster_index,
                   cluster_num, den_queue, center, clt_sem, batch_size, para_f, nv_flag):
    """cluster.cu:16-119.  ``l1_norm`` and ``index_mapper`` only drive the reference's sort prefilter
    (binary.cu:49-69) and are not needed by the grid-hash search; they are accepted and ignored.
    ``radius``/``min_pts`` are the 18-vectors of pbnet_ops.py:33-36 and must be uniform.
    Outputs are written in place; ``center`` and ``clt_sem`` are resized (cluster.cu:112-118)."""
    dev = _dev()
    r = radius.reshape(-1).to(torch.float32).cpu()
    m = min_pts.reshape(-1).to(torch.int32).cpu()
    if not (bool((r == r[0]).all()) and bool((m == m[0]).all())):
        raise ValueError("PB_lib.binary_cluster: per-class radius/min_pts tables must be uniform (pbnet_ops.py:33-36)")
    off = torch.stack([x, y, z], dim=1).to(dev, torch.float32)
    org = torch.stack([xo, yo, zo], dim=1).to(dev, torch.float32)
    sem_d = sem.to(dev, torch.int32)
    seg = batch_index.reshape(-1)[:batch_size].to(dev, torch.int32)
    uniform = True if sem_d.numel() == 0 else bool((sem_d == sem_d[0]).all().item())
    res = _ops.cluster_device(off, org, sem_d, seg, float(r[0]), int(m[0]), para_f=float(para_f),
                              nv_flag=bool(nv_flag), general_sem=not uniform, want_members=False)
    c = int(res.n_clusters.item())
    if c < 0:
        raise RuntimeError("PB_lib.binary_cluster: segment lengths do not sum to the point count, or class id "
                           "outside [2,19]")
    cluster_index.copy_(res.cluster_id)
    cluster_num[:batch_size].copy_(res.cluster_num)
    den_queue.copy_(res.den)
    center.resize_(3 * c)
    center.copy_(res.centers[:3 * c])
    clt_sem.resize_(c)
    clt_sem.copy_(res.clt_sem[:c])


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, nInstance, nProposal):
    """get_iou.cpp:9-18 -- writes ``proposals_iou`` [nProposal, nInstance] in place."""
    N.require_cuda(proposals_iou)
    iou = _ops.get_iou_device(proposals_idx.cuda(), proposals_offset.cuda(), instance_labels.cuda(),
                              instance_pointnum.cuda())
    proposals_iou.copy_(iou.view_as(proposals_iou))


def cal_iou_and_masklabel(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou,
                          nInstance, nProposal, mask_scores_sigmoid, mask_label, mode):
    """cal_iou_and_masklabel.cpp -- writes ``proposals_iou`` and ``mask_label`` in place."""
    N.require_cuda(proposals_iou, mask_label)
    idx = proposals_idx.to(torch.int32).contiguous()
    off = proposals_offset.to(torch.int32).contiguous()
    lab = instance_labels.to(torch.int64).contiguous()
    pnum = instance_pointnum.to(torch.int32).contiguous()
    ms = mask_scores_sigmoid.to(torch.float32).contiguous()
    assert proposals_iou.is_contiguous() and mask_label.is_contiguous()
    rc = N.lib().pbn_cal_iou_and_masklabel(N.ptr(idx), N.ptr(off), N.ptr(lab), N.ptr(pnum), N.ptr(proposals_iou),
                                           int(nInstance), int(nProposal), N.ptr(ms), N.ptr(mask_label), int(mode),
                                           N.current_stream())
    N.check(rc, "pbn_cal_iou_and_masklabel")


def cal_normal_line(xyz, face, normal_line, num_vtx, num_face):
    """normal/cal_normal.h:10 -- offline mesh preprocessing, unused by the reference's own pipeline
    (decode_scannet.py:113-117 commented out); outside the hot path."""
    raise NotImplementedError("cal_normal_line is outside the MI355X hot path")
