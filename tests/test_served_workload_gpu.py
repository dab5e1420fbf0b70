"""GPU tier: the `served` leg of bench.py at its size and dtype -- four DISTINCT configs[1] scenes (161-180 k points each, bf16
slabs) through the serving front (pbnet_amd/serving.py: one merged forward of four, n_batch = 4) against their four single
forwards.  In fp32 the merged forward equals the single ones (tests/test_batched_gpu.py); in bf16 a merged level may take another
kernel family / tile shape than the same level of a single scene, so the 16-bit contract of tests/test_bench_workload_gpu.py
applies: the same semantic predictions, the same NUMBER of proposals per scene, proposal membership differing on at most
BF16_MEMBERSHIP_BOUND of the rows, scores within BF16_SCORE_TOL; and the merged run is bit-identical run to run."""
import numpy as np
import pytest
import torch

from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet
from pbnet_amd.serving import SceneServer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_MEMBERSHIP_BOUND = 1e-3      # fraction of a scene's proposal rows (tests/test_bench_workload_gpu.py)
BF16_SCORE_TOL = 3e-3


def _sets(res):
    idx, off = res["proposals"][0].cpu().numpy(), res["proposals"][1].cpu().numpy()
    return [frozenset(idx[off[p]:off[p + 1], 1].tolist()) for p in range(len(off) - 1)]


def test_four_bench_scenes_through_the_serving_front_bf16():
    cfg = get_config(test=True)
    torch.manual_seed(22)                         # bench.py:build_model
    model = PBNet(cfg).to(DEV).eval()
    scenes = []
    for sd in (2, 3, 4, 5):
        b, t, info = synth.make_val_batch(seed=sd, copies=1)
        sc = {k: torch.from_numpy(b[k]).to(DEV) for k in ("xyz_voxel", "feat_voxel", "xyz_original", "v2p_index")}
        sc["feat_voxel"] = sc["feat_voxel"].to(torch.bfloat16)
        scenes.append((sc, {k: torch.from_numpy(v).to(DEV) for k, v in t.items()}, info["n_points"]))
    assert scenes[0][2] == 161517
    singles = []
    with torch.no_grad():
        for sc, t, _ in scenes:
            singles.append(model(sc["feat_voxel"], sc["xyz_voxel"], sc["xyz_original"], sc["v2p_index"], None, 1, "test", teacher=t, n_batch=1))
    runs = []
    for _ in range(2):
        srv = SceneServer(model, max_batch=4, forwards_in_flight=1)
        futs = [srv.submit(sc, t) for sc, t, _ in scenes]
        runs.append([f.result(timeout=600) for f in futs])
        assert srv.forwards == 1 and srv.scenes == 4, "the four waiting scenes were merged into one forward"
        srv.close()
    for a, b in zip(*runs):                        # run to run bit-identical
        assert torch.equal(a["proposals"][0], b["proposals"][0]) and torch.equal(a["clt_scores"], b["clt_scores"])
    for j, (got, want) in enumerate(zip(runs[0], singles)):
        assert torch.equal(got["sem_pred_p"], want["sem_pred_p"])
        gs, ws = _sets(got), _sets({"proposals": want["proposals"]})
        assert len(gs) == len(ws) and len(ws) >= 8, "scene %d: %d proposals against %d" % (j, len(gs), len(ws))
        rows = sum(len(s) for s in ws)
        diff = sum(len(a ^ b) for a, b in zip(gs, ws))          # same local scenes in the same order: compare position by position
        e_sc = (got["clt_scores"].float() - want["clt_scores"].float()).abs().max().item()
        print("scene %d (%d points): %d proposals, %d of %d rows differ (%.4f %%), scores max |diff| %.2e" % (
            j, scenes[j][2], len(ws), diff, rows, 100.0 * diff / rows, e_sc))
        assert diff / rows <= BF16_MEMBERSHIP_BOUND
        assert e_sc <= BF16_SCORE_TOL
