"""oracle/loss_ref.py -- TEST INFRASTRUCTURE ONLY: numpy (float64 accumulation of float32 terms) restatement of the loss
arithmetic of /root/reference/network/PBNet.py:366-416 (model_fn) and :463-468 (diceLoss), written from the formulas,
without torch's loss classes.  Never imported by the product path.

PARITY: partly pinned.  `get_segmented_scores` (tools/mIOU.py:34-49) is pinned by tests/golden/segmented_scores.npz
(outputs of the reference's own function, tests/golden/make_post_golden.py); `get_iou` by oracle/pb_cluster_ref.c
(get_iou.cu:12-29).  model_fn itself cannot be imported here (network/PBNet.py imports MinkowskiEngine at module
level), so the remaining terms are restated line by line:

  :374-375  semantic loss   = mean over rows with label != -100 of  -log softmax(score)[label]
  :380-384  offset L1 loss  = sum_i valid_i * |pred_i - gt_i|_1 / (sum valid + 1e-6),  gt = inst_info[:, :3] - xyz
  :387-392  direction loss  = sum_i valid_i * -(gt_i/(|gt_i|+1e-8) . pred_i/(|pred_i|+1e-8)) / (sum valid + 1e-6)
  :398-403  mask BCE        = mean over ALL rows of  w_i * -(t_i log p_i + (1-t_i) log(1-p_i)),  w = (gt != -1);
                              log clamped at -100 (torch.nn.BCELoss); t_i = 0 on ignore rows: the reference assigns 0.5
                              IN PLACE into a LONG tensor, which stores 0
  :405      dice            = 1 - (2 sum t p + 1) / (sum t^2 + sum p^2 + 1 + 1e-8) over EVERY row (gt_mask no longer
                              holds -1 after the in-place assignment, so `gt_mask != -1` selects all rows)
  :409-415  score BCE       = mean_p -(s_p log c_p + (1-s_p) log(1-c_p)),  s = segmented(max_i IoU[p, i])
"""
import numpy as np


def _log(x):
    with np.errstate(divide="ignore"):
        return np.maximum(np.log(x), -100.0)


def segmented_scores(scores, fg, bg):
    """tools/mIOU.py:34-49, elementwise."""
    s = np.asarray(scores, dtype=np.float32)
    out = np.zeros_like(s)
    k = np.float32(1.0 / (fg - bg))
    b = np.float32(bg / (bg - fg))
    for i, v in enumerate(s):
        if v > fg:
            out[i] = 1.0
        elif v < bg:
            out[i] = 0.0
        else:
            out[i] = np.float32(v * k + b)
    return out


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum):
    """get_iou.cu:12-29: iou[p, i] = inter / (len_p + num_i - inter + 1e-5)."""
    P, I = len(proposals_offset) - 1, len(instance_pointnum)
    out = np.zeros((P, I), np.float32)
    for p in range(P):
        pts = proposals_idx[proposals_offset[p]:proposals_offset[p + 1]]
        lab = instance_labels[pts]
        for i in range(I):
            inter = int((lab == i).sum())
            out[p, i] = np.float32(inter) / np.float32(float(len(pts) + int(instance_pointnum[i]) - inter) + 1e-5)
    return out


def losses(sem_score, offset_pred, sem_label, ins_label, inst_info, xyz, mask=None, proposals=None, clt_scores=None,
           instance_pointnum=None, fg=0.95, bg=0.20):
    """All numpy inputs.  mask = (pred_mask [R], gt_mask i64 [R] with -1 = ignore) or None; proposals =
    (point_idx i64 [S], offset i64 [P+1]).  Returns a dict of python floats (and the mutated gt_mask)."""
    f8 = np.float64
    sc = np.asarray(sem_score, np.float32).astype(f8)
    keep = sem_label != -100
    m = sc.max(1, keepdims=True)
    lse = (m + np.log(np.exp(sc - m).sum(1, keepdims=True)))[:, 0]
    nll = lse - sc[np.arange(len(sc)), np.where(keep, sem_label, 0)]
    out = {"semantic_loss": float(nll[keep].sum() / max(int(keep.sum()), 1))}
    gt = (np.asarray(inst_info, np.float32)[:, :3] - np.asarray(xyz, np.float32)).astype(np.float32)
    pr = np.asarray(offset_pred, np.float32)
    valid = (ins_label != -100).astype(np.float32)
    denom = f8(valid.sum(dtype=np.float32)) + 1e-6
    out["offset_norm_loss"] = float((np.abs(pr - gt).sum(1).astype(f8) * valid).sum() / denom)
    gn = np.sqrt((gt.astype(f8) ** 2).sum(1))
    pn = np.sqrt((pr.astype(f8) ** 2).sum(1))
    cos = ((gt / (gn[:, None] + 1e-8)) * (pr / (pn[:, None] + 1e-8))).sum(1)
    out["offset_dir_loss"] = float((-cos * valid).sum() / denom)
    out["loss"] = out["semantic_loss"] + out["offset_norm_loss"] + out["offset_dir_loss"]
    if mask is not None:
        p, g = np.asarray(mask[0], np.float32).astype(f8).reshape(-1), np.asarray(mask[1]).astype(np.int64).copy()
        w = (g != -1).astype(f8)
        g[g == -1] = 0                                   # int(0.5) == 0: the in-place assignment into a long tensor
        t = g.astype(f8)
        bce = -(t * _log(p) + (1.0 - t) * _log(1.0 - p)) * w
        out["mask_loss"] = float(bce.sum() / max(len(p), 1))
        out["dice_loss"] = float(1.0 - (2.0 * (t * p).sum() + 1.0) / ((t ** 2).sum() + (p ** 2).sum() + 1.0 + 1e-8))
        iou = get_iou(np.asarray(proposals[0]), np.asarray(proposals[1]), np.asarray(ins_label), np.asarray(instance_pointnum))
        s = segmented_scores(iou.max(1), fg, bg).astype(f8)
        c = np.asarray(clt_scores, np.float32).astype(f8).reshape(-1)
        out["score_loss"] = float((-(s * _log(c) + (1.0 - s) * _log(1.0 - c))).sum() / max(len(c), 1))
        out["loss"] += out["mask_loss"] + out["dice_loss"] + out["score_loss"]
        out["gt_mask"] = g
        out["gt_scores"] = s
    return out
