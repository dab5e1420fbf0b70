"""CPU tier: the C oracle against the committed golden fixtures and the independent brute force."""
import glob
import os

import numpy as np
import pytest

from oracle import pb_cluster_ref as oracle
import bruteforce_cluster as brute

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cluster_*.npz")))
KEYS = ("cluster_id", "cluster_num", "den_queue", "clt_sem")


def _load(path):
    return dict(np.load(path))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[8:-4] for p in GOLDEN])
def test_oracle_matches_golden(path):
    g = _load(path)
    res = oracle.binary_cluster(g["off"], g["org"], g["sem"], g["seg"], float(g["radius"]), int(g["min_pts"]),
                                nv_flag=bool(g["nv_flag"]))
    for k in KEYS:
        assert np.array_equal(res[k], g[k]), k
    assert np.array_equal(res["center"].view(np.int32), g["center"].view(np.int32))


@pytest.mark.parametrize("path", [p for p in GOLDEN if "G9" not in p],
                         ids=[os.path.basename(p)[8:-4] for p in GOLDEN if "G9" not in p])
def test_bruteforce_matches_golden(path):
    g = _load(path)
    res = brute.binary_cluster(g["off"], g["org"], g["sem"], g["seg"], float(g["radius"]), int(g["min_pts"]),
                               nv_flag=bool(g["nv_flag"]))
    for k in KEYS:
        assert np.array_equal(res[k], g[k]), k
    assert np.array_equal(res["center"].view(np.int32), g["center"].view(np.int32))


def test_designed_properties(golden_dir):
    # G3: the border LP (index 0) is LP, both plates are HP components, LP takes the LARGER id (=1, plate A).
    g = _load(os.path.join(golden_dir, "cluster_G3.npz"))
    assert g["den_queue"][0] < 31 and (g["den_queue"][1:] >= 31).all()
    assert g["cluster_id"][0] == 1 and set(g["cluster_id"][1:65]) == {0} and set(g["cluster_id"][65:]) == {1}
    # G4: clusters of 105 and 60 points dropped, 106 kept (threshold exactly 106.0, strict <)
    g = _load(os.path.join(golden_dir, "cluster_G4.npz"))
    assert g["cluster_num"].tolist() == [3]
    g2 = _load(os.path.join(golden_dir, "cluster_G4n.npz"))
    assert (g2["cluster_id"][120:225] == -1).all() and (g2["cluster_id"][225:331] == 1).all()
    assert (g2["cluster_id"][331:391] == -1).all() and (g2["cluster_id"][391:] == 2).all()
    # G5: ties resolved to the highest index
    g = _load(os.path.join(golden_dir, "cluster_G5.npz"))
    assert g["cluster_id"][40] == g["cluster_id"][162 - 1]  # 4-way tie -> last assigned point (cluster b)
    # G6: empty segment and a segment with no surviving cluster
    g = _load(os.path.join(golden_dir, "cluster_G6.npz"))
    assert g["cluster_num"].tolist() == [2, 0, 0, 1] and g["center"].shape == (3, 3)
    assert (g["cluster_id"][160:210] == -1).all() and g["cluster_id"][210:].max() == 2


def test_pbnet_ops_cluster_signature():
    g = _load(GOLDEN[0])
    cid, cnum, den, ctr = oracle.cluster(g["off"], g["org"], g["sem"], g["seg"], float(g["radius"]), int(g["min_pts"]))
    assert np.array_equal(den, g["den_queue"] + 1)          # pbnet_ops.py:75
    assert ctr.ndim == 1 and ctr.shape[0] == 3 * int(cnum.sum())


def test_get_iou_oracle():
    rng = np.random.default_rng(0)
    n, n_inst = 500, 7
    labels = rng.integers(-1, n_inst, n).astype(np.int64)
    labels[labels < 0] = -100
    pointnum = np.array([(labels == i).sum() for i in range(n_inst)], np.int32)
    offs = np.array([0, 50, 50, 180, 300], np.int32)
    idx = rng.integers(0, n, offs[-1]).astype(np.int32)
    iou = oracle.get_iou(idx, offs, labels, pointnum)
    for p in range(4):
        sel = labels[idx[offs[p]:offs[p + 1]]]
        for i in range(n_inst):
            inter = int((sel == i).sum())
            want = np.float32(np.float32(inter) / (np.float64(np.float32(len(sel) + int(pointnum[i]) - inter)) + 1e-5))
            assert iou[p, i] == want


def test_exact_fast_division_evidence():
    """k_centers replaces the IEEE division on its serial critical path by a reciprocal + FMA correction; the C check
    compares it bit for bit with the IEEE quotient (reduced range here; the full run is 640 M comparisons)."""
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(here, "oracle"), "fastdiv_check"])
    out = subprocess.run([os.path.join(here, "oracle", "fastdiv_check"), "30000", "120"], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout
