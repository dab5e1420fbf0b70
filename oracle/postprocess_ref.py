"""CPU restatement (numpy) of the evaluation-time post-processing that follows PBNet.forward
(/root/reference/eval_map.py:55-123, tools/mIOU.py:77-87, tools/getins.py:72-98).  TEST INFRASTRUCTURE ONLY: imported by
tests/ and by tests/golden/make_post_golden.py, never by the product.

Pinned: tests/golden/post_*.npz hold outputs of the reference's OWN `non_max_suppression` and `align_superpoint_label`
(imported from /root/reference/tools in the build container by tests/golden/make_post_golden.py) wrapped in the
tensor statements of eval_map.py:55-123; tests/test_oracle_post.py checks this restatement against them."""
import numpy as np

SEMANTIC_LABEL_IDX = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])  # eval_map.py:32


def non_max_suppression(ious, scores, threshold):
    """tools/mIOU.py:77-87, statement for statement."""
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


def align_superpoint_label(labels, superpoint, num_label, ignore_label=-100):
    """tools/getins.py:72-98 without scipy: label histogram per superpoint, first arg-max, bucket num_label = ignore."""
    col = labels.copy()
    col[col < 0] = num_label
    n_sp = len(np.unique(superpoint))
    label_map = np.zeros((n_sp, num_label + 1), np.float64)
    np.add.at(label_map, (superpoint, col), 1.0)
    label = np.argmax(label_map, axis=1).astype(np.int64)
    label[label == num_label] = ignore_label
    return label, label_map


def refine_instances(pred_sem, proposals_idx, proposals_offset, clt_score, point_num, superpoint, score_thresh,
                     npoint_thresh, nms_thresh):
    """eval_map.py:55-123 for one scene.  Returns dict(clusters i32[C, N/3], cluster_scores f32[C],
    cluster_semantic_id i64[C], plus the intermediates the tests compare)."""
    n_fold = point_num // 3
    n_prop = proposals_offset.shape[0] - 1
    first = proposals_idx[:, 1][proposals_offset[:-1]]
    semantic_id = SEMANTIC_LABEL_IDX[pred_sem[first]]                                   # :63-65
    folded = proposals_idx[:, 1] % (point_num / 3)                                       # :67 (float modulus of ints)
    proposals_pred = np.zeros((n_prop, n_fold), np.int32)                                # :68-70
    proposals_pred[proposals_idx[:, 0], folded.astype(np.int64)] = 1
    pointnum_all = proposals_pred.sum(1)
    score_mask = clt_score > np.float32(score_thresh)                                    # :74
    clt = clt_score[score_mask]
    pp = proposals_pred[score_mask]
    sid = semantic_id[score_mask]
    npoint_mask = pp.sum(1) > npoint_thresh                                              # :80-81
    clt, pp, sid = clt[npoint_mask], pp[npoint_mask], sid[npoint_mask]
    out = dict(pointnum=pointnum_all.astype(np.int32))
    if sid.shape[0] == 0:                                                                # :87-88
        out.update(clusters=np.zeros((0, n_fold), np.int32), cluster_scores=np.zeros(0, np.float32),
                   cluster_semantic_id=np.zeros(0, np.int64), cross_ious=np.zeros((0, 0), np.float32),
                   pick=np.zeros(0, np.int32))
        return out
    ppf = pp.astype(np.float32)                                                          # :90-96
    inter = ppf @ ppf.T
    pn = ppf.sum(1)
    cross_ious = inter / (pn[:, None] + pn[None, :] - inter)
    pick = non_max_suppression(cross_ious, clt, nms_thresh)                              # :97-98
    clusters = pp[pick]
    cluster_scores = clt[pick]
    cluster_sid = sid[pick]
    seg = np.full(n_fold, -100, np.int64)                                                # :104-108
    for c_i in range(clusters.shape[0]):
        seg[clusters[c_i] == 1] = c_i
    sp_labels, _ = align_superpoint_label(seg, superpoint, clusters.shape[0])            # :109
    seg2 = sp_labels[superpoint]                                                         # :110
    clusters = np.zeros_like(clusters)                                                   # :112-118
    keep = []
    for c_i in range(clusters.shape[0]):
        cur = seg2 == c_i
        if cur.any():
            keep.append(c_i)
        clusters[c_i, cur] = 1
    keep = np.asarray(keep, np.int64)
    out.update(clusters=clusters[keep], cluster_scores=cluster_scores[keep], cluster_semantic_id=cluster_sid[keep],
               cross_ious=cross_ious.astype(np.float32), pick=pick, seg=seg, seg_refined=seg2)
    return out
