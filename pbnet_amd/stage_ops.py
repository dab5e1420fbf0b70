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
    N.require_cuda(packed, member_idx, ins_ind, xyz, point_feat, sem_score)
    dev = point_feat.device
    c = int(point_feat.shape[1])
    ld_out = c + 2 if ld_out is None else int(ld_out)
    assert packed.dtype == torch.int32 and packed.numel() == 4 * n_ent + 1 and packed.is_contiguous()
    assert point_feat.stride(1) == 1 and sem_score.stride(1) == 1 and sem_score.dtype == point_feat.dtype
    assert xyz.dtype == torch.float32 and xyz.is_contiguous() and ins_ind.dtype == torch.int64
    assert member_idx.dtype == torch.int32 and (sem_pred is None or sem_pred.dtype == torch.int64)
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


class _HeadParams(object):
    """fp32 device copies of one two-layer head (Linear, BatchNorm(eval), PReLU, Linear[, Sigmoid]) in the layout
    pbn_mlp_rows reads; rebuilt when any parameter / running statistic changes."""

    def __init__(self):
        self.key = None

    def get(self, head):
        lin1, bn, act, lin2 = head[0].linear, head[1].bn, head[2].module, head[3].linear
        tensors = [lin1.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, act.weight, lin2.weight, lin2.bias]
        key = tuple((t.data_ptr(), t._version) for t in tensors if t is not None)
        if key != self.key:
            with torch.no_grad():
                f = lambda t: t.detach().float().contiguous()
                scale = f(bn.weight) / torch.sqrt(f(bn.running_var) + bn.eps)
                shift = f(bn.bias) - f(bn.running_mean) * scale
                hidden = lin1.weight.shape[0]
                self.w1 = f(lin1.weight)
                self.scale, self.shift = scale.contiguous(), shift.contiguous()
                self.slope = f(act.weight).expand(hidden).contiguous()
                self.w2 = f(lin2.weight)
                self.b2 = f(lin2.bias) if lin2.bias is not None else None
                self.hidden, self.n_out, self.channels = int(hidden), int(lin2.weight.shape[0]), int(lin1.weight.shape[1])
                self.sigmoid = len(head) > 4
                assert lin1.bias is None
            self.key = key
        return self


_HEADS = {}


def mlp_rows(head, feats, idx_a=None, idx_b=None, n=None):
    """head(x) for the rows feats[idx_b[idx_a[i]]] (PBNet.py:43-82 heads, eval mode) in one launch; returns [n, n_out]."""
    hp = _HEADS.setdefault(id(head), _HeadParams()).get(head)
    N.require_cuda(feats)
    assert feats.stride(1) == 1 and feats.shape[1] == hp.channels
    if n is None:
        n = int(idx_a.shape[0]) if idx_a is not None else int(feats.shape[0])
    out = torch.empty(n, hp.n_out, dtype=feats.dtype, device=feats.device)
    vp = N.c_vp
    rc = N.lib().pbn_mlp_rows(vp(feats.data_ptr()), feats.stride(0), hp.channels, N.ptr(idx_a), N.ptr(idx_b), int(n),
                              N.ptr(hp.w1), N.ptr(hp.scale), N.ptr(hp.shift), N.ptr(hp.slope), hp.hidden, N.ptr(hp.w2),
                              N.ptr(hp.b2), hp.n_out, int(hp.sigmoid), vp(out.data_ptr()), hp.n_out, _DT[feats.dtype],
                              N.current_stream())
    N.check(rc, "pbn_mlp_rows")
    return out


def sem_argmax_table(score, batch, nb):
    """PBNet.py:134,151-163: (sem_pred i64[N], sem_prob [N] own-class softmax score, table i32[S, nb], block_hist)."""
    N.require_cuda(score)
    n, s = int(score.shape[0]), int(score.shape[1])
    dev = score.device
    assert score.stride(1) == 1 and (batch is None or (batch.dtype == torch.int32 and batch.is_contiguous()))
    lib = N.lib()
    sem_pred = torch.empty(n, dtype=torch.int64, device=dev)
    sem_prob = torch.empty(n, dtype=score.dtype, device=dev)
    table = torch.empty(s, nb, dtype=torch.int32, device=dev)
    block_hist = torch.empty(max(lib.pbn_select_blocks(n), 1), s, dtype=torch.int32, device=dev)
    rc = lib.pbn_sem_argmax_table(N.c_vp(score.data_ptr()), score.stride(0), s, N.ptr(batch), int(nb), n, _DT[score.dtype],
                                  N.ptr(sem_pred), N.c_vp(sem_prob.data_ptr()), N.ptr(table), N.ptr(block_hist),
                                  N.current_stream())
    N.check(rc, "pbn_sem_argmax_table")
    return sem_pred, sem_prob, table, block_hist


def select_points(sem_pred, class_base, block_hist, xyz, offset, m):
    """PBNet.py:151-170: class-major stable selection; class_base i32[S] on the device (-1 drops a class), m = number of
    selected points (host).  Returns (ins_ind i64[m], ins_orig f32[m,3], ins_offseted f32[m,3], ins_sem i32[m])."""
    N.require_cuda(sem_pred, class_base, block_hist, xyz, offset)
    dev = xyz.device
    n, s = int(sem_pred.shape[0]), int(class_base.shape[0])
    assert xyz.dtype == torch.float32 and xyz.is_contiguous() and offset.stride(1) == 1 and class_base.dtype == torch.int32
    ins_ind = torch.empty(m, dtype=torch.int64, device=dev)
    ins_orig = torch.empty(m, 3, dtype=torch.float32, device=dev)
    ins_off = torch.empty(m, 3, dtype=torch.float32, device=dev)
    ins_sem = torch.empty(m, dtype=torch.int32, device=dev)
    rc = N.lib().pbn_select_points(N.ptr(sem_pred), n, s, N.ptr(class_base), N.ptr(block_hist), N.ptr(xyz),
                                   N.c_vp(offset.data_ptr()), offset.stride(0), _DT[offset.dtype], N.ptr(ins_ind),
                                   N.ptr(ins_orig), N.ptr(ins_off), N.ptr(ins_sem), N.current_stream())
    N.check(rc, "pbn_select_points")
    return ins_ind, ins_orig, ins_off, ins_sem


def mask_count(mask_score, thd, row_scene, n_scenes):
    """First half of get_proposal (PBNet.py:317-333): kept rows per local scene and per row block (device tensors)."""
    N.require_cuda(mask_score, row_scene)
    n = int(row_scene.shape[0])
    dev = mask_score.device
    lib = N.lib()
    ms = mask_score.view(n, -1)
    assert ms.stride(1) == 1 and row_scene.dtype == torch.int64
    per_scene = torch.empty(max(n_scenes, 1), dtype=torch.int32, device=dev)
    block_cnt = torch.empty(max(lib.pbn_select_blocks(n), 1), dtype=torch.int32, device=dev)
    rc = lib.pbn_mask_count(N.c_vp(ms.data_ptr()), ms.stride(0), float(thd), N.ptr(row_scene), n, int(n_scenes),
                            _DT[ms.dtype], N.ptr(per_scene), N.ptr(block_cnt), N.current_stream())
    N.check(rc, "pbn_mask_count")
    return per_scene[:n_scenes], block_cnt


def proposal_rows(mask_score, thd, row_scene, point_idx, dense_of, block_cnt, total, xyz=None, scale=1.0, voxel=1.0,
                  point_feat=None):
    """Second half of get_proposal (+ PBNet.py:240-252 when xyz / point_feat are given).  Returns (proposals_idx i64[P,2],
    proposals_ms [P], coords i32[P,4] or None, feat [P,C] or None)."""
    n = int(row_scene.shape[0])
    dev = mask_score.device
    ms = mask_score.view(n, -1)
    prop_idx = torch.empty(total, 2, dtype=torch.int64, device=dev)
    prop_ms = torch.empty(total, dtype=ms.dtype, device=dev)
    coords = torch.empty(total, 4, dtype=torch.int32, device=dev) if xyz is not None else None
    feat = None
    c = ld_feat = 0
    if point_feat is not None:
        assert point_feat.stride(1) == 1 and point_feat.dtype == ms.dtype
        c, ld_feat = int(point_feat.shape[1]), point_feat.stride(0)
        feat = torch.empty(total, c, dtype=point_feat.dtype, device=dev)
    if total == 0:               # no row passed the threshold: nothing to launch (empty tensors have no address)
        return prop_idx, prop_ms, coords, feat
    vp = N.c_vp
    rc = N.lib().pbn_proposal_rows(
        vp(ms.data_ptr()), ms.stride(0), float(thd), N.ptr(row_scene), N.ptr(point_idx), n, N.ptr(dense_of),
        N.ptr(block_cnt), N.ptr(xyz), float(np.float32(scale)), reciprocal_f32(voxel),
        None if point_feat is None else vp(point_feat.data_ptr()), ld_feat, c, _DT[ms.dtype], N.ptr(prop_idx),
        vp(prop_ms.data_ptr()), N.ptr(coords), None if feat is None else vp(feat.data_ptr()), N.current_stream())
    N.check(rc, "pbn_proposal_rows")
    return prop_idx, prop_ms, coords, feat
