"""Python face of csrc/stages.hip: the fused glue between the big kernels of PBNet.forward on the inference path
(/root/reference/network/PBNet.py:113-280).  Device tensors in, device tensors out, no host synchronisation; every
function is one launch.  (With autograd enabled PBNet.forward keeps the differentiable tensor-op form.)"""
import numpy as np
import torch

from . import _native as N

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def reciprocal_f32(x):
    """fp32 reciprocal of a host scalar: what a device tensor divided by a host scalar is multiplied by."""
    return float(np.float32(1.0) / np.float32(x))


def local_scene_rows(packed, n_ent, n_rows, member_idx, ins_ind, xyz, voxel, point_feat, sem_score, sem_pred, ld_out=None):
    """PBNet.py:182-247 in one launch.  packed i32[4*n_ent+1] on the device = [row_start (n_ent+1) | member_start |
    scene | weight bits].  Returns (point_idx i64[R], row_scene i64[R], coords i32[R,4], feat [R, C+2])."""
    N.require_cuda(packed, member_idx, ins_ind, xyz, point_feat, sem_score, sem_pred)
    dev = point_feat.device
    c = int(point_feat.shape[1])
    ld_out = c + 2 if ld_out is None else int(ld_out)
    assert packed.dtype == torch.int32 and packed.numel() == 4 * n_ent + 1 and packed.is_contiguous()
    assert point_feat.stride(1) == 1 and sem_score.stride(1) == 1 and sem_score.dtype == point_feat.dtype
    assert xyz.dtype == torch.float32 and xyz.is_contiguous() and ins_ind.dtype == torch.int64
    assert member_idx.dtype == torch.int32 and sem_pred.dtype == torch.int64
    point_idx = torch.empty(n_rows, dtype=torch.int64, device=dev)
    row_scene = torch.empty(n_rows, dtype=torch.int64, device=dev)
    coords = torch.empty(n_rows, 4, dtype=torch.int32, device=dev)
    feat = torch.empty(n_rows, ld_out, dtype=point_feat.dtype, device=dev)
    base, isz = packed.data_ptr(), 4
    vp = N.c_vp
    rc = N.lib().pbn_local_scene_rows(
        vp(base), vp(base + isz * (n_ent + 1)), vp(base + isz * (2 * n_ent + 1)), vp(base + isz * (3 * n_ent + 1)),
        int(n_ent), int(n_rows), N.ptr(member_idx), N.ptr(ins_ind), N.ptr(xyz), reciprocal_f32(voxel),
        vp(point_feat.data_ptr()), point_feat.stride(0), c, vp(sem_score.data_ptr()), sem_score.stride(0),
        N.ptr(sem_pred), _DT[point_feat.dtype], N.ptr(point_idx), N.ptr(row_scene), N.ptr(coords),
        vp(feat.data_ptr()), ld_out, N.current_stream())
    N.check(rc, "pbn_local_scene_rows")
    return point_idx, row_scene, coords, feat
