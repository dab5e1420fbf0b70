#!/usr/bin/env python3
"""Golden vectors for the post-processing row, produced IN THE BUILD CONTAINER by the reference's
own functions: `tools.mIOU.non_max_suppression` and `tools.getins.align_superpoint_label` are imported from
/root/reference and wrapped in the tensor statements of eval_map.py:55-123 (that code lives inside eval_epoch and cannot
be imported as a function; the statements below follow it line by line on CPU tensors).

    python tests/golden/make_post_golden.py         # writes tests/golden/post_P*.npz

Inputs are synthetic: overlapping proposals over a 3-copy batch, scores with ties and near-threshold values, a proposal
that dies at the score gate, one at the point-count gate, one that vanishes in the superpoint vote."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from tools.mIOU import non_max_suppression          # noqa: E402  (reference code, executed here only)
from tools.getins import align_superpoint_label     # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEMANTIC_LABEL_IDX = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]


def make_case(seed, n_fold, n_prop, n_sp):
    rng = np.random.default_rng(seed)
    point_num = 3 * n_fold
    centers = rng.integers(0, n_fold, n_prop)
    idx_rows = []
    for p in range(n_prop):
        size = int(rng.integers(40, 900))
        if p == 1:
            size = 60                                   # dies at the point-count gate
        lo = max(0, centers[p] - size // 2)
        base = np.arange(lo, min(n_fold, lo + size))
        base = base[rng.random(base.shape[0]) < 0.8]
        if p % 4 == 0 and p > 0:                        # heavy overlap with the previous proposal
            base = np.union1d(base, idx_rows[-1][1][: len(idx_rows[-1][1]) // 2] % n_fold)
        copy = rng.integers(0, 3, base.shape[0])
        pts = np.unique(base + copy * n_fold)
        pts = np.concatenate([pts, pts[: len(pts) // 7] % n_fold + ((copy[: len(pts) // 7] + 1) % 3) * n_fold])  # fold duplicates
        idx_rows.append((np.full(pts.shape[0], p), pts))
    proposals_idx = np.stack([np.concatenate([r[0] for r in idx_rows]), np.concatenate([r[1] for r in idx_rows])], 1).astype(np.int64)
    lens = np.array([r[0].shape[0] for r in idx_rows])
    proposals_offset = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    clt = rng.random(n_prop).astype(np.float32)
    clt[0] = 0.01                                       # dies at the score gate
    clt[3] = clt[2]                                     # a tie
    pred_sem = rng.integers(0, 20, point_num).astype(np.int64)
    # compressed superpoint ids: contiguous runs of 20-200 points, every id present
    sp = np.repeat(np.arange(n_sp), rng.multinomial(n_fold - n_sp, np.ones(n_sp) / n_sp) + 1)[:n_fold].astype(np.int64)
    sp = sp[rng.permutation(n_fold)] if seed == 1 else sp     # P1: scattered superpoints -> every cluster loses the vote
    return dict(point_num=point_num, proposals_idx=proposals_idx, proposals_offset=proposals_offset, clt=clt,
                pred_sem=pred_sem, superpoint=sp)


def reference_flow(c, score_t, npoint_t, nms_t):
    """eval_map.py:55-123 on CPU tensors, calling the reference's two functions."""
    point_num = c["point_num"]
    proposals_idx = torch.from_numpy(c["proposals_idx"].copy())
    proposals_offset = torch.from_numpy(c["proposals_offset"])
    clt_score = torch.from_numpy(c["clt"])
    pred_sem = torch.from_numpy(c["pred_sem"])
    superpoint = torch.from_numpy(c["superpoint"])
    semantic_id = torch.tensor(SEMANTIC_LABEL_IDX)
    test = pred_sem[proposals_idx[:, 1][proposals_offset[:-1].long()].long()]
    semantic_id = semantic_id[test]
    proposals_idx[:, 1] = proposals_idx[:, 1] % (point_num / 3)
    proposals_pred = torch.zeros((proposals_offset.shape[0] - 1, point_num // 3), dtype=torch.int)
    proposals_pred[proposals_idx[:, 0].long(), proposals_idx[:, 1].long()] = 1
    pointnum_all = proposals_pred.sum(1).numpy().astype(np.int32)
    score_mask = (clt_score > score_t)
    clt_score = clt_score[score_mask]
    proposals_pred = proposals_pred[score_mask]
    semantic_id = semantic_id[score_mask]
    proposals_pointnum = proposals_pred.sum(1)
    npoint_mask = (proposals_pointnum > npoint_t)
    clt_score = clt_score[npoint_mask]
    proposals_pred = proposals_pred[npoint_mask]
    semantic_id = semantic_id[npoint_mask]
    proposals_pred_f = proposals_pred.float()
    intersection = torch.mm(proposals_pred_f, proposals_pred_f.t())
    proposals_pointnum = proposals_pred_f.sum(1)
    proposals_pn_h = proposals_pointnum.unsqueeze(-1).repeat(1, proposals_pointnum.shape[0])
    proposals_pn_v = proposals_pointnum.unsqueeze(0).repeat(proposals_pointnum.shape[0], 1)
    cross_ious = intersection / (proposals_pn_h + proposals_pn_v - intersection)
    pick_idxs = non_max_suppression(cross_ious.numpy(), clt_score.numpy(), nms_t)
    clusters = proposals_pred[pick_idxs]
    cluster_scores = clt_score[pick_idxs]
    cluster_semantic_id = semantic_id[pick_idxs]
    seg_result = torch.ones(point_num // 3) * -100
    for c_i in range(clusters.shape[0]):
        cur_idx = torch.nonzero(clusters[c_i, :] == 1).view(-1)
        seg_result[cur_idx] = c_i
    seg_result = seg_result.type(torch.int64)
    seg0 = seg_result.clone().numpy()
    sp_labels, sp_scores = align_superpoint_label(seg_result, superpoint, clusters.shape[0])
    seg_result = sp_labels[superpoint]
    clusters[:, :] = 0
    pick2 = [p_i for p_i in range(clusters.shape[0])]
    for c_i in range(clusters.shape[0]):
        cur_idx = torch.nonzero(seg_result == c_i).view(-1)
        if cur_idx.shape[0] == 0:
            pick2.remove(c_i)
        clusters[c_i, cur_idx] = 1
    clusters = clusters[pick2]
    cluster_scores = cluster_scores[pick2]
    cluster_semantic_id = cluster_semantic_id[pick2]
    return dict(pointnum=pointnum_all, cross_ious=cross_ious.numpy(), pick=np.asarray(pick_idxs, np.int32), seg=seg0,
                seg_refined=seg_result.numpy(), clusters=clusters.numpy().astype(np.int32),
                cluster_scores=cluster_scores.numpy(), cluster_semantic_id=cluster_semantic_id.numpy())


def main():
    cases = {"P1": (1, 3000, 12, 60), "P2": (2, 20011, 40, 700), "P3": (3, 54001, 25, 1500), "P4": (4, 997, 6, 11)}
    for name, (seed, n_fold, n_prop, n_sp) in cases.items():
        c = make_case(seed, n_fold, n_prop, n_sp)
        ref = reference_flow(c, 0.07, 101, 0.10)
        np.savez_compressed(os.path.join(HERE, "post_%s.npz" % name), score_t=0.07, npoint_t=101, nms_t=0.10,
                            **{"in_" + k: v for k, v in c.items()}, **{"out_" + k: v for k, v in ref.items()})
        print(name, "proposals", n_prop, "->", ref["clusters"].shape[0], "clusters; picked", len(ref["pick"]))


if __name__ == "__main__":
    main()


def segmented_scores_golden():
    """tools/mIOU.py:34-49 (used by model_fn, network/PBNet.py:412): outputs of the reference function itself."""
    from tools.mIOU import get_segmented_scores
    rng = np.random.default_rng(7)
    s = rng.random(4096).astype(np.float32)
    s[:6] = [0.0, 0.25, 0.75, 1.0, 0.2499999, 0.7500001]
    out = {}
    for fg, bg in ((0.75, 0.25), (1.0, 0.0), (0.5, 0.2)):
        out["fg%g_bg%g" % (fg, bg)] = get_segmented_scores(torch.from_numpy(s), fg, bg).numpy()
    np.savez_compressed(os.path.join(HERE, "segmented_scores.npz"), scores=s, **out)
    print("segmented_scores: %d thresholds" % len(out))


if __name__ == "__main__":
    segmented_scores_golden()
