"""CPU: the evaluator row.  (1) The oracle restatement against the goldens produced by the reference's
own tools/eval.py (tests/golden/make_eval_golden.py); (2) the product's host half (pbnet_amd/evaluate.py: matching, AP
integration, averages, formats) on association tables taken from the same goldens -- no device call on this side."""
import glob
import os

import numpy as np
import pytest

from oracle import evaluate_ref as O
from pbnet_amd import evaluate as E

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "eval_E*.npz")))


def _scenes(g):
    return range(int(g["n_scenes"]))


def _sorted_rows(a):
    a = np.asarray(a).reshape(-1, a.shape[-1])
    return a[np.lexsort(a.T[::-1])] if a.shape[0] else a


def _same_ap(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0))


def test_goldens_present():
    assert len(CASES) == 3


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference(path):
    g = np.load(path)
    tables = []
    for s in _scenes(g):
        t = O.assign(g["s%d_conf" % s], g["s%d_label" % s], g["s%d_mask" % s], g["s%d_gt" % s])
        assert np.array_equal(t[0], g["s%d_gt_rows" % s])
        assert np.array_equal(t[1], g["s%d_pred_rows" % s])
        assert np.array_equal(t[2].view(np.int32), g["s%d_pred_conf" % s].view(np.int32))
        assert np.array_equal(_sorted_rows(t[3]), _sorted_rows(g["s%d_pairs" % s]))
        tables.append(t)
    ap = O.evaluate(tables)
    assert ap.dtype == np.float32 and _same_ap(ap, g["ap"])                       # bit-equal, nan pattern included
    head, per = O.averages(ap)
    assert np.allclose(head, g["avg"], rtol=0, atol=1e-6, equal_nan=True)
    assert np.allclose(per, g["class_avg"], rtol=0, atol=1e-6, equal_nan=True)


def _record(g, s):
    """SceneMatches from a golden's flat tables (what the device association produces, tests/test_eval_gpu.py)."""
    gt, pr, conf, pairs = g["s%d_gt_rows" % s], g["s%d_pred_rows" % s], g["s%d_pred_conf" % s], g["s%d_pairs" % s]
    go, po = np.argsort(gt[:, 1], kind="stable"), np.argsort(pr[:, 1], kind="stable")
    gt, pr, conf = gt[go], pr[po], conf[po]
    inter = np.zeros((pr.shape[0], gt.shape[0]), np.int64)
    col = {int(u): j for j, u in enumerate(gt[:, 1])}
    for p, u, c in pairs:
        inter[int(p), col[int(u)]] = c
    label_id = np.array(O.VALID_CLASS_IDS, np.int64)[pr[:, 0]]
    return E.SceneMatches("scene%04d_00" % s, gt[:, 0], gt[:, 1], gt[:, 2], pr[:, 0], pr[:, 1], label_id, pr[:, 2], pr[:, 3],
                          conf, inter)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_product_ap_matches_reference(path):
    g = np.load(path)
    matches = {"scene%04d_00" % s: _record(g, s) for s in _scenes(g)}
    ap = E.evaluate_matches(matches)
    assert ap.dtype == np.float32 and ap.shape == (1, 18, 10) and _same_ap(ap, g["ap"])
    avgs = E.compute_averages(ap)
    got = np.array([avgs["all_ap"], avgs["all_ap_50%"], avgs["all_ap_25%"]], np.float64)
    assert np.array_equal(got, g["avg"], equal_nan=True)                          # same float32 reductions -> same bits
    per = np.array([[avgs["classes"][n][k] for k in ("ap", "ap50%", "ap25%")] for n in E.CLASS_LABELS], np.float64)
    assert np.array_equal(per, g["class_avg"], equal_nan=True)
    lines = E.format_results(avgs)
    assert len(lines) == 4 + 18 + 3 and lines[2].startswith("what") and "average" in lines[-2]


@pytest.mark.parametrize("path", CASES[:2], ids=[os.path.basename(p)[:-4] for p in CASES[:2]])
def test_reference_layout_round_trip(path):
    """to_reference() is the dict layout tools/eval.py builds; from_reference() of it gives the same AP -- so a caller may
    mix this evaluator with the reference's functions on either side."""
    g = np.load(path)
    matches = {}
    for s in _scenes(g):
        rec = _record(g, s)
        gt2pred, pred2gt = rec.to_reference()
        assert set(gt2pred) == set(E.CLASS_LABELS) == set(pred2gt)
        flat = [(li, d["instance_id"], d["vert_count"]) for li, n in enumerate(E.CLASS_LABELS) for d in gt2pred[n]]
        assert np.array_equal(np.array(flat, np.int64).reshape(-1, 3), g["s%d_gt_rows" % s])
        flat = [(li, d["pred_id"], d["vert_count"], d["void_intersection"]) for li, n in enumerate(E.CLASS_LABELS)
                for d in pred2gt[n]]
        assert np.array_equal(np.array(flat, np.int64).reshape(-1, 4), g["s%d_pred_rows" % s])
        pairs = [(p["pred_id"], d["instance_id"], p["intersection"]) for n in E.CLASS_LABELS for d in gt2pred[n]
                 for p in d["matched_pred"]]
        assert np.array_equal(np.array(pairs, np.int64).reshape(-1, 3), g["s%d_pairs" % s])     # same order, too
        for n in E.CLASS_LABELS:
            for d in pred2gt[n]:
                assert d["filename"] == "%s_%03d" % (rec.scene, d["pred_id"])
        matches[rec.scene] = dict(gt=gt2pred, pred=pred2gt)
    assert _same_ap(E.evaluate_matches(matches), g["ap"])


def test_gt_id_encoding_and_file_format(tmp_path):
    rng = np.random.default_rng(5)
    n = 4000
    ins = np.full(n, -100, np.int64)
    sem = np.full(n, -100, np.int64)
    cuts = np.sort(rng.choice(np.arange(1, n), 12, replace=False))
    for j in range(6):
        ins[cuts[2 * j]:cuts[2 * j + 1]] = j
        sem[cuts[2 * j]:cuts[2 * j + 1]] = int(rng.integers(0, 20))
    sem[cuts[2]:cuts[3]] = -100                                # an instance over unlabelled points -> class 0 (wall id 1)
    perm = rng.permutation(n)
    ins, sem = ins[perm], sem[perm]
    want = O.encode_gt_ids(sem, ins)
    got = E.encode_gt_ids(sem.astype(np.float64), ins.astype(np.float64))   # the .npy files hold floats (decode_scannet.py)
    assert got.dtype == np.int32 and np.array_equal(got, want)
    assert (got[ins < 0] == 0).all() and set(np.unique(got[ins == 1]) // 1000) == {1}
    path = tmp_path / "scene0000_00.txt"
    E.save_gt_ids(path, got)
    assert np.array_equal(E.load_gt_ids(path), got)
    assert path.read_text().splitlines()[0] == str(int(got[0]))
    assert np.array_equal(E.encode_gt_ids(np.zeros(0), np.zeros(0)), np.zeros(0, np.int32))
    assert np.array_equal(E.encode_gt_ids(np.full(5, -100), np.full(5, -100)), np.zeros(5, np.int32))
