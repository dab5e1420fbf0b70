"""Evaluation-time post-processing after PBNet.forward -- the MI355X form of /root/reference/eval_map.py:55-123
(TTA fold, score / point-count thresholds, mask-IoU NMS, superpoint alignment), SURVEY.md 8(f) rank 1.

`refine_instances` takes exactly what eval_map.py has in hand at line 55 (the `pred` dict of model_fn_eval, the number of
points of the 3-copy batch and the scene's superpoint ids) and returns what it holds at line 118:
(clusters i32[C, N/3], cluster_scores [C], cluster_semantic_id i64[C]) on the device.  Masks are bitsets on the device
(csrc/post.hip); the greedy NMS runs on the host on a [P, P] matrix with the reference's own numpy statements."""
import numpy as np
import torch

from . import _native as N

SEMANTIC_LABEL_IDX = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]   # eval_map.py:32


def non_max_suppression(ious, scores, threshold):
    """tools/mIOU.py:77-87 (host, numpy): greedy NMS over a few hundred proposals."""
    ixs = scores.argsort()[::-1]
    pick = []
    while len(ixs) > 0:
        i = ixs[0]
        pick.append(i)
        iou = ious[i, ixs[1:]]
        remove_ixs = np.where(iou > threshold)[0] + 1
        ixs = np.delete(ixs, remove_ixs)
        ixs = np.delete(ixs, 0)
    return np.array(pick, dtype=np.int32)


def refine_instances(pred_sem, proposals, clt_scores, point_num, superpoint, cfg, return_debug=False):
    proposals_idx, proposals_offset = proposals[0], proposals[1]
    N.require_cuda(proposals_idx, proposals_offset, pred_sem)
    dev = proposals_idx.device
    lib = N.lib()
    st = N.current_stream()
    n_fold = int(point_num) // 3
    n_prop = int(proposals_offset.shape[0]) - 1
    words = lib.pbn_post_words(n_fold)
    clt_score = clt_scores.view(-1).float()
    empty = (torch.zeros(0, n_fold, dtype=torch.int32, device=dev), clt_score[:0], torch.zeros(0, dtype=torch.int64, device=dev))
    if n_prop <= 0:
        return empty
    # eval_map.py:63-65: class of a proposal = class of its first member
    label_idx = torch.tensor(SEMANTIC_LABEL_IDX, device=dev)
    semantic_id = label_idx[pred_sem[proposals_idx[:, 1][proposals_offset[:-1].long()].long()]]
    # :67-70 + :80: folded bitsets and their sizes
    pidx = proposals_idx.contiguous()
    masks = torch.empty(n_prop, words, dtype=torch.int32, device=dev)
    counts = torch.empty(n_prop, dtype=torch.int32, device=dev)
    N.check(lib.pbn_proposal_bitmask(N.ptr(pidx), int(pidx.shape[0]), n_fold, n_prop, N.ptr(masks), N.ptr(counts), st),
            "pbn_proposal_bitmask")
    # :74-84 thresholds (host: P scalars)
    host = torch.cat([clt_score, counts.float()]).cpu().numpy()
    score_h, count_h = host[:n_prop], host[n_prop:].astype(np.int64)
    rows = np.nonzero(score_h > np.float32(cfg.TEST_SCORE_THRESH))[0]
    rows = rows[count_h[rows] > cfg.TEST_NPOINT_THRESH]
    if rows.shape[0] == 0:
        return empty
    # :90-98 mask IoU of the survivors + greedy NMS
    rows_d = torch.from_numpy(rows.astype(np.int32)).to(dev)
    r = int(rows.shape[0])
    iou = torch.empty(r, r, dtype=torch.float32, device=dev)
    N.check(lib.pbn_mask_iou(N.ptr(masks), N.ptr(rows_d), r, n_fold, N.ptr(counts), N.ptr(iou), st), "pbn_mask_iou")
    pick = non_max_suppression(iou.cpu().numpy(), score_h[rows], cfg.TEST_NMS_THRESH)
    pick_rows = rows[pick]
    n_pick = int(pick_rows.shape[0])
    # :104-116 superpoint alignment and rebuilt clusters
    sp = torch.as_tensor(superpoint).to(dev).long().contiguous()
    n_sp = int(sp.max().item()) + 1
    pick_d = torch.from_numpy(pick_rows.astype(np.int32)).to(dev)
    seg = torch.empty(n_fold, dtype=torch.int64, device=dev)
    seg2 = torch.empty(n_fold, dtype=torch.int64, device=dev)
    hist = torch.empty(n_sp, n_pick + 1, dtype=torch.int32, device=dev)
    sp_label = torch.empty(n_sp, dtype=torch.int64, device=dev)
    masks2 = torch.empty(n_pick, words, dtype=torch.int32, device=dev)
    counts2 = torch.empty(n_pick, dtype=torch.int32, device=dev)
    N.check(lib.pbn_superpoint_refine(N.ptr(masks), N.ptr(pick_d), n_pick, n_fold, N.ptr(sp), n_sp, N.ptr(seg), N.ptr(hist),
                                      N.ptr(sp_label), N.ptr(seg2), N.ptr(masks2), N.ptr(counts2), st),
            "pbn_superpoint_refine")
    keep = np.nonzero(counts2.cpu().numpy() > 0)[0]                                   # :113-118 drop vanished clusters
    keep_d = torch.from_numpy(keep.astype(np.int32)).to(dev)
    clusters = torch.empty(int(keep.shape[0]), n_fold, dtype=torch.int32, device=dev)
    N.check(lib.pbn_bitmask_to_dense(N.ptr(masks2), N.ptr(keep_d), int(keep.shape[0]), n_fold, N.ptr(clusters), st),
            "pbn_bitmask_to_dense")
    sel = torch.from_numpy(pick_rows[keep].astype(np.int64)).to(dev)
    out = (clusters, clt_score[sel], semantic_id[sel])
    if return_debug:
        return out + (dict(pointnum=counts, cross_ious=iou, pick=pick, seg=seg, seg_refined=seg2),)
    return out
