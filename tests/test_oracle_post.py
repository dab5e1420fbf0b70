"""CPU tier: the post-processing oracle (oracle/postprocess_ref.py) against golden vectors produced by the reference's
own `non_max_suppression` / `align_superpoint_label` (tests/golden/make_post_golden.py).  This row IS pinned."""
import glob
import os

import numpy as np
import pytest

from oracle import postprocess_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "post_P*.npz")))


def test_goldens_present():
    assert len(CASES) >= 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_outputs(path):
    g = np.load(path)
    out = R.refine_instances(g["in_pred_sem"], g["in_proposals_idx"], g["in_proposals_offset"], g["in_clt"],
                             int(g["in_point_num"]), g["in_superpoint"], float(g["score_t"]), int(g["npoint_t"]),
                             float(g["nms_t"]))
    assert np.array_equal(out["pointnum"], g["out_pointnum"])
    assert np.array_equal(out["cross_ious"], g["out_cross_ious"])          # same fp32 quotients of exact integers
    assert np.array_equal(out["pick"], g["out_pick"])
    assert np.array_equal(out["seg"], g["out_seg"])
    assert np.array_equal(out["seg_refined"], g["out_seg_refined"])
    assert np.array_equal(out["clusters"], g["out_clusters"])
    assert np.array_equal(out["cluster_scores"], g["out_cluster_scores"])
    assert np.array_equal(out["cluster_semantic_id"], g["out_cluster_semantic_id"])


def test_get_segmented_scores_matches_reference_function():
    """network/PBNet.py:412 -> tools/mIOU.py:34-49; golden = outputs of the reference function (make_post_golden.py)."""
    import torch
    from pbnet_amd.network.PBNet import get_segmented_scores
    g = np.load(os.path.join(HERE, "golden", "segmented_scores.npz"))
    s = torch.from_numpy(g["scores"])
    for key in g.files:
        if key == "scores":
            continue
        fg, bg = (float(v[2:]) for v in key.split("_"))
        assert np.array_equal(get_segmented_scores(s, fg, bg).numpy(), g[key]), key
