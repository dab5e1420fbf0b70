"""GPU tier: the evaluator's association step (pbn_instance_overlap through pbnet_amd/evaluate.py) against the golden
vectors of the reference's own tools/eval.py and, at full scene size, against the oracle's brute-force statement.
Integers bit-exact; the AP tensor bit-equal."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import evaluate_ref as O
from pbnet_amd import evaluate as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "eval_E*.npz")))


def _flat(rec):
    """SceneMatches -> the oracle's / golden's flat tables (class-major rows, pairs sorted)."""
    go = np.lexsort((rec.gt_id, rec.gt_class))
    po = np.lexsort((rec.pred_id, rec.pred_class))
    gt_rows = np.stack([rec.gt_class, rec.gt_id, rec.gt_vert], 1)[go].reshape(-1, 3)
    pred_rows = np.stack([rec.pred_class, rec.pred_id, rec.pred_vert, rec.pred_void], 1)[po].reshape(-1, 4)
    q, g = np.nonzero(rec.inter)
    pairs = np.stack([rec.pred_id[q], rec.gt_id[g], rec.inter[q, g]], 1).reshape(-1, 3)
    return gt_rows, pred_rows, np.asarray(rec.pred_conf, np.float32)[po], _sorted_rows(pairs)


def _sorted_rows(a):
    return a[np.lexsort(a.T[::-1])] if a.shape[0] else a


def _same_ap(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_association_and_ap_match_reference(path):
    g = np.load(path)
    matches = {}
    for s in range(int(g["n_scenes"])):
        pred = dict(conf=g["s%d_conf" % s], label_id=g["s%d_label" % s], mask=g["s%d_mask" % s])
        rec = E.assign_instances_for_scan("scene%04d_00" % s, pred, g["s%d_gt" % s], device=DEV)
        gt_rows, pred_rows, conf, pairs = _flat(rec)
        assert np.array_equal(gt_rows, g["s%d_gt_rows" % s])
        assert np.array_equal(pred_rows, g["s%d_pred_rows" % s])
        assert np.array_equal(conf.view(np.int32), g["s%d_pred_conf" % s].view(np.int32))
        assert np.array_equal(pairs, _sorted_rows(g["s%d_pairs" % s]))
        matches[rec.scene] = rec
    ap = E.evaluate_matches(matches)
    assert _same_ap(ap, g["ap"])
    avgs = E.compute_averages(ap)
    assert np.array_equal(np.array([avgs["all_ap"], avgs["all_ap_50%"], avgs["all_ap_25%"]], np.float64), g["avg"],
                          equal_nan=True)


def _big_scene(seed, n_pts, n_inst, n_pred):
    rng = np.random.default_rng(seed)
    gt = np.zeros(n_pts, np.int64)
    cuts = np.sort(rng.choice(np.arange(1, n_pts), 2 * n_inst, replace=False))
    for j in range(n_inst):
        cls = [1, 2, 3, 4, 5, 7, 39][int(rng.integers(0, 7))]
        gt[cuts[2 * j]:cuts[2 * j + 1]] = cls * 1000 + j + 1
    perm = rng.permutation(n_pts)                              # vertices of an instance are not contiguous in a mesh
    scatter = rng.random() < 0.5
    masks = np.zeros((n_pred, n_pts), np.int32)
    label = np.zeros(n_pred, np.int64)
    for p in range(n_pred):
        j = int(rng.integers(0, n_inst))
        lo, hi = int(cuts[2 * j]), int(cuts[2 * j + 1])
        a = max(0, lo - int(rng.integers(0, 400)))
        b = min(n_pts, hi + int(rng.integers(-200, 400)))
        masks[p, a:max(a + 1, b)] = int(rng.integers(1, 5))
        masks[p, rng.integers(0, n_pts, 50)] = 1               # stray points anywhere
        label[p] = [3, 4, 5, 7, 39, 13][int(rng.integers(0, 6))]
    if scatter:
        gt, masks = gt[perm], masks[:, perm]
    return gt, dict(conf=rng.random(n_pred).astype(np.float32), label_id=label, mask=masks)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_full_size_scene_against_oracle(seed):
    gt, pred = _big_scene(seed, 161517, 60, 48)
    want = O.assign(pred["conf"], pred["label_id"], pred["mask"], gt)
    dev_pred = dict(pred, mask=torch.from_numpy(pred["mask"]).to(DEV))           # eval_map.py hands over a device tensor
    rec = E.assign_instances_for_scan("big", dev_pred, gt)
    got = _flat(rec)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2].view(np.int32), want[2].view(np.int32))
    assert np.array_equal(got[3], _sorted_rows(want[3]))
    assert _same_ap(E.evaluate_matches([rec]), O.evaluate([want]))
    # properties that hold at any size: row sums = mask sizes, column sums bounded by instance sizes, total conserved
    inter, uid = E.overlap_table(dev_pred["mask"], gt)
    assert np.array_equal(inter.sum(1), (pred["mask"] != 0).sum(1))
    assert (inter <= np.bincount(np.searchsorted(uid, gt))[None, :]).all()


def test_mask_dtypes_and_many_ids():
    rng = np.random.default_rng(3)
    n = 50000
    gt = 3_000_000 + rng.permutation(n).astype(np.int64)      # one id per point: 50 000 bins, the no-LDS path
    gt[rng.random(n) < 0.3] = 0
    masks = (rng.random((5, n)) < 0.4)
    want = np.zeros((5, len(np.unique(gt))), np.int64)
    uid, idx = np.unique(gt, return_inverse=True)
    for p in range(5):
        want[p] = np.bincount(idx[masks[p]], minlength=uid.shape[0])
    for m in (masks, masks.astype(np.int8), masks.astype(np.int64) * 9, torch.from_numpy(masks).to(DEV),
              torch.from_numpy(masks.astype(np.float32)).to(DEV)):
        inter, u = E.overlap_table(m, gt, device=DEV)
        assert np.array_equal(u, uid) and np.array_equal(inter, want)


def test_edge_cases(tmp_path):
    gt = np.zeros(1000, np.int64)
    gt[100:400] = 3001
    gt[500:560] = 4002                                         # below the region size: ignored instance
    path = tmp_path / "scene.txt"
    E.save_gt_ids(path, gt)
    empty = dict(conf=np.zeros(0, np.float32), label_id=np.zeros(0, np.int64), mask=np.zeros((0, 1000), np.int64))
    rec = E.assign_instances_for_scan("s", empty, str(path), device=DEV)
    assert rec.pred_id.shape[0] == 0 and rec.inter.shape == (0, 2) and list(rec.gt_vert) == [300, 60]
    ap = E.evaluate_matches({"s": rec})
    assert ap[0, 0, 0] == 0.0 and np.isnan(ap[0, 1, 0]) and np.isnan(ap[0, 5, 0])   # gt without predictions / nothing
    m = np.zeros((2, 1000), np.int32)
    m[0, 90:410] = 1
    m[1, 480:600] = 1                                          # sits on the ignored instance + void
    rec = E.assign_instances_for_scan("s", dict(conf=np.array([0.9, 0.8], np.float32), label_id=np.array([3, 4]), mask=m),
                                      path, device=DEV)
    assert list(rec.pred_vert) == [320, 120] and list(rec.pred_void) == [20, 60] and rec.inter.tolist() == [[300, 0], [0, 60]]
    ap = E.evaluate_matches({"s": rec})
    assert ap[0, 0, 0] == 1.0 and np.isnan(ap[0, 1, 0])        # class 4 has no countable ground truth
    with pytest.raises(ValueError):
        E.assign_instances_for_scan("s", dict(conf=np.zeros(1), label_id=np.array([3]), mask=np.zeros((1, 999))), gt, device=DEV)
    none = np.zeros(500, np.int64)                             # a scene without any annotation
    rec = E.assign_instances_for_scan("n", dict(conf=np.array([0.5], np.float32), label_id=np.array([3]),
                                                mask=np.ones((1, 500), np.int32)), none, device=DEV)
    assert rec.gt_id.shape[0] == 0 and list(rec.pred_void) == [500]
    assert np.isnan(E.evaluate_matches([rec])).all()


def test_ground_truth_as_prediction_scores_one():
    """Whole evaluator chain on a bench-sized synthetic scene: ids from (semantic, instance) labels (get_val_gt.py
    encoding), the ground-truth instances themselves as predictions -> AP 1 at every threshold for every class present,
    nan elsewhere; the same masks cut to their first half (IoU exactly 1/2) fail the strict `> 0.5` test but pass 0.25."""
    from pbnet_amd import synth
    batch, teacher, _ = synth.make_val_batch(seed=2, copies=1)
    ins = batch["ins"].astype(np.int64)
    sem = teacher["sem_score"].argmax(1).astype(np.int64)
    for i in np.unique(ins[ins >= 0]):                       # one class per instance, as in a real scan
        sem[ins == i] = sem[np.nonzero(ins == i)[0][0]]
    gt = E.encode_gt_ids(sem, ins)
    assert np.array_equal(gt, O.encode_gt_ids(sem, ins))
    ids = [int(i) for i in np.unique(gt) if i >= 1000 and (i // 1000) in E.VALID_CLASS_IDS]
    big = [i for i in ids if (gt == i).sum() >= E.MIN_REGION_SIZE]
    assert len(big) >= 5
    masks = np.stack([(gt == i).astype(np.int32) for i in ids])
    pred = dict(conf=np.linspace(0.9, 0.5, len(ids)).astype(np.float32), label_id=np.array([i // 1000 for i in ids]),
                mask=torch.from_numpy(masks).to(DEV))
    ap = E.evaluate_matches({"s": E.assign_instances_for_scan("s", pred, gt)})
    present = sorted({E._CLASS_OF_ID[i // 1000] for i in big})
    for li in range(len(E.CLASS_LABELS)):
        if li in present:
            assert (ap[0, li] == 1.0).all(), li
        else:
            assert np.isnan(ap[0, li]).all(), li
    avgs = E.compute_averages(ap)
    assert avgs["all_ap"] == avgs["all_ap_50%"] == avgs["all_ap_25%"] == 1.0
    half = masks.copy()
    for r, i in enumerate(ids):
        pts = np.nonzero(gt == i)[0]
        half[r, pts[len(pts) // 2:]] = 0                      # keeps floor(n/2) points: IoU <= 1/2
    ap2 = E.evaluate_matches({"s": E.assign_instances_for_scan("s", dict(pred, mask=half), gt, device=DEV)})
    o25 = int(np.argmin(np.abs(E.OVERLAPS - 0.25)))
    kept = [li for li in present if any((gt == i).sum() // 2 >= E.MIN_REGION_SIZE for i in big if E._CLASS_OF_ID[i // 1000] == li)]
    for li in kept:
        assert ap2[0, li, 0] == 0.0 and ap2[0, li, o25] > 0.0, li
