#!/usr/bin/env python3
"""Golden vectors for the ScanNet AP evaluator row, produced IN THE BUILD CONTAINER by the
reference's own `tools.eval` (`assign_instances_for_scan`, `evaluate_matches`, `compute_averages`), imported from
/root/reference and run on synthetic scenes.

    python tests/golden/make_eval_golden.py         # writes tests/golden/eval_E*.npz

Each case is a small "validation set": per scene a ground-truth id vector in the val_gt encoding
(`datasets/scannetv2/get_val_gt.py:26-39`: 0 = unannotated, class_id * 1000 + instance + 1) and a prediction triple
(conf, label_id, mask) as eval_map.py:128-131 builds it.  Stored: the inputs, the flattened association tables the
reference produced (so the device overlap kernel is pinned entry by entry) and the AP tensor / averages.

Cases cover: predictions over void and wall/floor points, instances below the 100-vertex region size, predictions
below it, several predictions on one instance (the lower-scored one turns false positive), confidence ties, a class
with ground truth but no prediction (AP 0) and the reverse, classes absent from both (nan), a label id outside the
benchmark classes, an empty prediction list, a scene without annotated instances."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, "/root/reference")
import tools.eval as ref_eval          # noqa: E402  (reference code, executed here only)

HERE = os.path.dirname(os.path.abspath(__file__))
VALID = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]


def make_scene(rng, n_pts, n_inst, n_pred, classes, flavour):
    """Instances are contiguous index runs (what a mask looks like does not matter to the evaluator, only counts)."""
    gt = np.zeros(n_pts, np.int64)
    cuts = np.sort(rng.choice(np.arange(1, n_pts), 2 * n_inst, replace=False))
    spans = []
    for j in range(n_inst):
        lo, hi = int(cuts[2 * j]), int(cuts[2 * j + 1])
        cls = int(classes[rng.integers(0, len(classes))])
        if flavour == "small" and j % 3 == 0:
            hi = min(hi, lo + int(rng.integers(20, 99)))            # below MIN_REGION_SIZES
        if j == 1:
            cls = 1                                                  # wall: annotated, not a benchmark class (void)
        if j == 2:
            cls = 2                                                  # floor
        gt[lo:hi] = cls * 1000 + j + 1
        spans.append((lo, hi, cls))
    conf, label, masks = [], [], []
    for p in range(n_pred):
        lo, hi, cls = spans[int(rng.integers(0, n_inst))]
        mode = p % 6
        m = np.zeros(n_pts, np.int64)
        if mode in (0, 1):                                           # good overlap, jittered ends
            a = max(0, lo + int(rng.integers(-30, 30)))
            b = min(n_pts, hi + int(rng.integers(-30, 30)))
            m[a:max(a + 1, b)] = 1
        elif mode == 2:                                              # partial overlap (passes only low thresholds)
            m[lo:lo + max(1, (hi - lo) * int(rng.integers(30, 70)) // 100)] = 1
        elif mode == 3:                                              # spills into the void next to the instance
            m[max(0, lo - (hi - lo)):hi] = 1
        elif mode == 4:                                              # wrong class on a real instance
            m[lo:hi] = 1
            cls = int(classes[rng.integers(0, len(classes))])
        else:                                                        # tiny prediction (dropped: < 100 vertices)
            m[lo:lo + int(rng.integers(5, 99))] = 7                  # any non-zero value counts as "in the mask"
        if cls in (1, 2):
            cls = int(classes[0])
        conf.append(np.float32(rng.integers(1, 20)) / np.float32(20.0) if flavour == "ties" else np.float32(rng.random()))
        label.append(cls)
        masks.append(m)
    if flavour == "odd_label" and n_pred:
        label[0] = 13                                                # not a ScanNet benchmark id: skipped
    pred = dict(conf=np.array(conf, np.float32), label_id=np.array(label, np.int64),
                mask=np.stack(masks) if masks else np.zeros((0, n_pts), np.int64))
    return gt, pred


def flatten(gt2pred, pred2gt):
    """The reference's nested dicts as flat int tables: gt rows (label, instance_id, vert_count), pred rows (label,
    pred_id, vert_count, void_intersection) with the confidence beside them, and (pred_id, instance_id, intersection)
    triples in the order the reference appended them to `matched_pred` of each instance."""
    gt_rows, pred_rows, pred_conf, pairs = [], [], [], []
    for li, name in enumerate(ref_eval.CLASS_LABELS):
        for g in gt2pred[name]:
            gt_rows.append((li, g["instance_id"], g["vert_count"]))
            for p in g["matched_pred"]:
                pairs.append((p["pred_id"], g["instance_id"], p["intersection"]))
        for p in pred2gt[name]:
            pred_rows.append((li, p["pred_id"], p["vert_count"], p["void_intersection"]))
            pred_conf.append(p["confidence"])
            for g in p["matched_gt"]:                                # cross-check: both directions hold the same pairs
                assert (p["pred_id"], g["instance_id"], g["intersection"]) in pairs
    return (np.array(gt_rows, np.int64).reshape(-1, 3), np.array(pred_rows, np.int64).reshape(-1, 4),
            np.array(pred_conf, np.float32), np.array(pairs, np.int64).reshape(-1, 3))


def run_case(name, seed, scenes):
    rng = np.random.default_rng(seed)
    out, matches = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for si, spec in enumerate(scenes):
            gt, pred = make_scene(rng, **spec)
            path = os.path.join(tmp, "s%d.txt" % si)
            np.savetxt(path, gt, fmt="%d")                            # the val_gt file format (get_val_gt.py:39)
            scene = "scene%04d_00" % si
            gt2pred, pred2gt = ref_eval.assign_instances_for_scan(scene, pred, path)
            matches[scene] = dict(gt=gt2pred, pred=pred2gt)
            g, p, c, pr = flatten(gt2pred, pred2gt)
            out.update({"s%d_gt" % si: gt, "s%d_conf" % si: pred["conf"], "s%d_label" % si: pred["label_id"],
                        "s%d_mask" % si: pred["mask"].astype(np.int8), "s%d_gt_rows" % si: g, "s%d_pred_rows" % si: p,
                        "s%d_pred_conf" % si: c, "s%d_pairs" % si: pr})
        ap = ref_eval.evaluate_matches(matches)
        avgs = ref_eval.compute_averages(ap)
    out["n_scenes"] = np.int64(len(scenes))
    out["ap"] = ap
    out["avg"] = np.array([avgs["all_ap"], avgs["all_ap_50%"], avgs["all_ap_25%"]], np.float64)
    out["class_avg"] = np.array([[avgs["classes"][n]["ap"], avgs["classes"][n]["ap50%"], avgs["classes"][n]["ap25%"]]
                                 for n in ref_eval.CLASS_LABELS], np.float64)
    np.savez_compressed(os.path.join(HERE, "eval_%s.npz" % name), **out)
    print(name, "ap shape", ap.shape, "all_ap %.4f ap50 %.4f ap25 %.4f" % tuple(out["avg"]),
          "nan classes", int(np.isnan(ap[0, :, 0]).sum()))


if __name__ == "__main__":
    few = VALID[:5]
    run_case("E1", 1, [dict(n_pts=6000, n_inst=8, n_pred=14, classes=few, flavour="plain"),
                       dict(n_pts=5000, n_inst=6, n_pred=10, classes=few, flavour="plain")])
    run_case("E2", 2, [dict(n_pts=8000, n_inst=12, n_pred=30, classes=VALID, flavour="small"),
                       dict(n_pts=7000, n_inst=9, n_pred=24, classes=VALID, flavour="ties"),
                       dict(n_pts=3000, n_inst=4, n_pred=0, classes=VALID, flavour="plain")])
    run_case("E3", 3, [dict(n_pts=9000, n_inst=10, n_pred=36, classes=VALID[:3], flavour="ties"),
                       dict(n_pts=4000, n_inst=5, n_pred=12, classes=VALID[2:6], flavour="odd_label"),
                       dict(n_pts=2500, n_inst=3, n_pred=6, classes=[1, 2], flavour="plain")])
