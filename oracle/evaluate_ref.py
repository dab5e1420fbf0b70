"""CPU restatement (numpy + plain loops) of the ScanNet AP evaluator the reference runs after every validation scene
(/root/reference/tools/eval.py:27-250, tools/getins.py:7-70, datasets/scannetv2/get_val_gt.py:26-39).
TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product.

Pinned: tests/golden/eval_E*.npz hold the outputs of the reference's OWN `tools.eval.assign_instances_for_scan`,
`evaluate_matches` and `compute_averages` (imported from /root/reference in the build container by
tests/golden/make_eval_golden.py); tests/test_oracle_eval.py checks this restatement against them.

The statement works on flat tables instead of the reference's nested dicts:
  gt_rows   int64[G, 3]  (class index, instance id, vertex count), class-major then ascending id
  pred_rows int64[Q, 4]  (class index, prediction id, vertex count, void intersection), class-major then id
  pred_conf f32[Q]
  pairs     int64[M, 3]  (prediction id, instance id, intersection) for intersection > 0 and equal classes"""
import numpy as np

VALID_CLASS_IDS = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]     # eval.py:10
N_CLASS = len(VALID_CLASS_IDS)
OVERLAPS = np.append(np.arange(0.5, 0.95, 0.05), 0.25)                                  # eval.py:18
MIN_REGION = 100                                                                        # eval.py:20
SEMANTIC_LABEL_IDX = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]


def encode_gt_ids(sem_label, ins_label):
    """get_val_gt.py:26-37, one instance at a time."""
    out = np.zeros(len(ins_label), np.int32)
    for inst in range(int(np.max(ins_label)) + 1 if len(ins_label) else 0):
        where = np.where(ins_label == inst)[0]
        if len(where) == 0:
            continue                      # (the reference indexes where[0] and raises; such scenes do not occur)
        sem = int(sem_label[where[0]])
        out[where] = SEMANTIC_LABEL_IDX[0 if sem == -100 else sem] * 1000 + inst + 1
    return out


def assign(pred_conf, pred_label, pred_mask, gt_ids):
    """eval.py:205-250 by brute force: one boolean pass per (prediction, instance) pair."""
    gt_ids = np.asarray(gt_ids)
    inst = [int(u) for u in np.unique(gt_ids) if u != 0 and int(u) // 1000 in VALID_CLASS_IDS]       # getins.py:59-70
    void = ~np.isin(gt_ids // 1000, VALID_CLASS_IDS)                                                 # eval.py:217
    gt_rows = [(VALID_CLASS_IDS.index(u // 1000), u, int((gt_ids == u).sum())) for u in inst]
    pred_rows, conf, pairs = [], [], []
    for i in range(len(pred_label)):
        label = int(pred_label[i])
        if label not in VALID_CLASS_IDS:
            continue
        inside = np.asarray(pred_mask[i]) != 0
        if inside.sum() < MIN_REGION:
            continue
        pid = len(pred_rows)
        pred_rows.append((VALID_CLASS_IDS.index(label), pid, int(inside.sum()), int((void & inside).sum())))
        conf.append(pred_conf[i])
        for u in inst:
            if u // 1000 == label:
                both = int(((gt_ids == u) & inside).sum())
                if both > 0:
                    pairs.append((pid, u, both))
    gt_rows.sort(key=lambda r: (r[0], r[1]))
    order = sorted(range(len(pred_rows)), key=lambda j: (pred_rows[j][0], pred_rows[j][1]))
    return (np.array(gt_rows, np.int64).reshape(-1, 3), np.array([pred_rows[j] for j in order], np.int64).reshape(-1, 4),
            np.array([conf[j] for j in order], np.float32), np.array(pairs, np.int64).reshape(-1, 3))


def _curve_area(y_true, y_score, hard_fn):
    """eval.py:131-176."""
    order = np.argsort(y_score)
    ys, yt = y_score[order], y_true[order]
    csum = np.cumsum(yt)
    _, firsts = np.unique(ys, return_index=True)
    total = csum[-1] if len(csum) else 0
    prec, rec = np.zeros(len(firsts) + 1), np.zeros(len(firsts) + 1)
    for k, f in enumerate(firsts):
        lower = csum[f - 1] if f > 0 else 0
        tp = total - lower
        fp = len(ys) - f - tp
        fn = lower + hard_fn
        prec[k], rec[k] = float(tp) / (tp + fp), float(tp) / (tp + fn)
    prec[-1], rec[-1] = 1.0, 0.0
    padded = np.concatenate([[rec[0]], rec, [0.0]])
    return np.dot(prec, np.convolve(padded, [-0.5, 0, 0.5], "valid"))


def evaluate(scenes):
    """eval.py:27-190 over a list of (gt_rows, pred_rows, pred_conf, pairs) tables -> ap f32[1, classes, overlaps]."""
    ap = np.zeros((1, N_CLASS, len(OVERLAPS)), np.float32)
    for oi, th in enumerate(OVERLAPS):
        visited = [set() for _ in scenes]
        for li in range(N_CLASS):
            y_true, y_score, hard_fn, has_gt, has_pred = [], [], 0, False, False
            for si, (gt_rows, pred_rows, conf, pairs) in enumerate(scenes):
                preds = {int(r[1]): (int(r[2]), int(r[3]), conf[j]) for j, r in enumerate(pred_rows) if r[0] == li}
                gts_all = {int(r[1]): int(r[2]) for r in gt_rows if r[0] == li}
                gts = [u for u in sorted(gts_all) if u >= 1000 and gts_all[u] >= MIN_REGION]
                has_gt |= len(gts) > 0
                has_pred |= len(preds) > 0
                mine = sorted((int(p), int(u), int(c)) for p, u, c in pairs if int(p) in preds)      # prediction order
                iou = {(p, u): c / (gts_all[u] + preds[p][0] - c) for p, u, c in mine}
                for u in gts:
                    score = None
                    for p, u2, _ in mine:
                        if u2 != u or p in visited[si] or not iou[(p, u)] > th:
                            continue
                        c = preds[p][2]
                        if score is None:
                            score = c
                            visited[si].add(p)
                        else:
                            y_true.append(0)
                            y_score.append(min(score, c))
                            score = max(score, c)
                    if score is None:
                        hard_fn += 1
                    else:
                        y_true.append(1)
                        y_score.append(score)
                for p in sorted(preds):
                    if any(iou[(q, u)] > th for q, u, _ in mine if q == p):
                        continue
                    ignore = preds[p][1] + sum(c for q, u, c in mine if q == p and (u < 1000 or gts_all[u] < MIN_REGION))
                    if float(ignore) / preds[p][0] <= th:
                        y_true.append(0)
                        y_score.append(preds[p][2])
            if has_gt and has_pred:
                ap[0, li, oi] = _curve_area(np.array(y_true, np.float64), np.array(y_score, np.float64), hard_fn)
            elif has_gt:
                ap[0, li, oi] = 0.0
            else:
                ap[0, li, oi] = np.nan
    return ap


def averages(ap):
    """eval.py:193-210 -> ((all_ap, ap50, ap25), per-class [classes, 3])."""
    is25, is50 = np.isclose(OVERLAPS, 0.25), np.isclose(OVERLAPS, 0.5)
    head = (np.nanmean(ap[0][:, ~is25]), np.nanmean(ap[0][:, is50]), np.nanmean(ap[0][:, is25]))
    per = np.stack([ap[0][:, ~is25].mean(1), ap[0][:, is50].mean(1), ap[0][:, is25].mean(1)], 1)
    return np.array(head, np.float64), per.astype(np.float64)
