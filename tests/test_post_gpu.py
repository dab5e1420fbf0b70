"""GPU tier: evaluation-time post-processing (csrc/post.hip through pbnet_amd/postprocess.py) against the golden vectors
of the reference's own functions (eval_map.py:55-123, tools/mIOU.py:77-87, tools/getins.py:72-98).  Bit-exact."""
import glob
import os
import types

import numpy as np
import pytest
import torch

from pbnet_amd import postprocess as PP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "post_P*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_refine_instances_matches_reference(path):
    g = np.load(path)
    cfg = types.SimpleNamespace(TEST_SCORE_THRESH=float(g["score_t"]), TEST_NPOINT_THRESH=int(g["npoint_t"]),
                                TEST_NMS_THRESH=float(g["nms_t"]))
    d = lambda k: torch.from_numpy(g[k]).to(DEV)
    proposals = (d("in_proposals_idx"), d("in_proposals_offset"), None, None)
    clusters, scores, sem_id, dbg = PP.refine_instances(d("in_pred_sem"), proposals, d("in_clt"), int(g["in_point_num"]),
                                                        g["in_superpoint"], cfg, return_debug=True)
    assert np.array_equal(dbg["pointnum"].cpu().numpy(), g["out_pointnum"])
    assert np.array_equal(dbg["cross_ious"].cpu().numpy(), g["out_cross_ious"])
    assert np.array_equal(dbg["pick"], g["out_pick"])
    assert np.array_equal(dbg["seg"].cpu().numpy(), g["out_seg"])
    assert np.array_equal(dbg["seg_refined"].cpu().numpy(), g["out_seg_refined"])
    assert np.array_equal(clusters.cpu().numpy(), g["out_clusters"])
    assert np.array_equal(scores.cpu().numpy(), g["out_cluster_scores"])
    assert np.array_equal(sem_id.cpu().numpy(), g["out_cluster_semantic_id"])


def test_no_survivors():
    cfg = types.SimpleNamespace(TEST_SCORE_THRESH=2.0, TEST_NPOINT_THRESH=101, TEST_NMS_THRESH=0.1)
    g = np.load(CASES[1])
    d = lambda k: torch.from_numpy(g[k]).to(DEV)
    clusters, scores, sem_id = PP.refine_instances(d("in_pred_sem"), (d("in_proposals_idx"), d("in_proposals_offset")),
                                                   d("in_clt"), int(g["in_point_num"]), g["in_superpoint"], cfg)
    assert clusters.shape[0] == 0 and scores.shape[0] == 0 and sem_id.shape[0] == 0
