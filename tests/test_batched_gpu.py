"""GPU tier: three DISTINCT scenes through one forward (the reference's own batch axis: network/PBNet.py:167-176 groups per
(class, batch element), dataset_preprocess.py:296 collates with a batch index) against the three single-scene forwards,
fp32: every proposal of the batched forward is a proposal of exactly one scene with the same point set (integers exact, point
indices shifted by the scene's offset) and the same score within 1e-4 -- batching changes launch shapes, not results.  Round 6:
the serving front built on that property (pbnet_amd/serving.py): eight scenes of DIFFERENT sizes submitted at once are served
by merged forwards (n_batch = scenes in the forward) and every scene gets back its own forward's result in the reference's form."""
import numpy as np
import pytest
import torch

from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCENE = dict(room=(1.6, 1.3, 1.2), n_boxes=6, pitch=0.03, classes=(17, 10))


def _forward(model, b, t):
    bd = {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}
    td = {k: torch.from_numpy(v).to(DEV) for k, v in t.items()}
    with torch.no_grad():
        return model(bd["feat_voxel"], bd["xyz_voxel"], bd["xyz_original"], bd["v2p_index"], None, 1, "test", teacher=td)


def _proposal_sets(ret, lo=0, hi=None):
    idx, off = ret["proposals"][0].cpu().numpy(), ret["proposals"][1].cpu().numpy()
    sc = ret["clt_scores"].float().cpu().numpy()
    out = {}
    for p in range(len(off) - 1):
        pts = idx[off[p]:off[p + 1], 1]
        if hi is not None and not (lo <= pts.min() and pts.max() < hi):
            continue
        out[frozenset((pts - lo).tolist())] = float(sc[p])
    return out


def test_three_scenes_in_one_forward_equal_three_forwards():
    cfg = get_config(test=True)
    torch.manual_seed(22)
    model = PBNet(cfg).to(DEV).eval()
    parts = [synth.make_val_batch(seed=s, copies=1, **SCENE) for s in (1, 2, 3)]
    singles = [_forward(model, b, t) for b, t, _ in parts]
    vox, feat, xyz, v2p, off, score, starts = [], [], [], [], [], [], [0]
    nv = 0
    for j, (b, t, _) in enumerate(parts):
        xv = b["xyz_voxel"].copy(); xv[:, 0] = j
        vox.append(xv); feat.append(b["feat_voxel"]); xyz.append(b["xyz_original"]); v2p.append(b["v2p_index"] + nv)
        off.append(t["offset"]); score.append(t["sem_score"])
        nv += len(xv); starts.append(starts[-1] + len(b["xyz_original"]))
    bb = dict(xyz_voxel=np.concatenate(vox), feat_voxel=np.concatenate(feat), xyz_original=np.concatenate(xyz),
              v2p_index=np.concatenate(v2p))
    tb = dict(sem_score=np.concatenate(score), offset=np.concatenate(off))
    batched = _forward(model, bb, tb)
    n_b = int(batched["proposals"][1].shape[0]) - 1
    n_s = [int(r["proposals"][1].shape[0]) - 1 for r in singles]
    print("proposals: batched %d, single scenes %s" % (n_b, n_s))
    assert n_b == sum(n_s) and min(n_s) >= 2
    for j, r in enumerate(singles):
        want = _proposal_sets(r)
        got = _proposal_sets(batched, starts[j], starts[j + 1])
        assert set(got) == set(want), "scene %d: proposal point sets" % j
        err = max(abs(got[k] - want[k]) for k in want)
        print("scene %d: %d proposals, scores max |diff| %.2e" % (j, len(want), err))
        assert err <= 1e-4
    # semantic predictions per point are the single scenes' (teacher forced here; the head path is row-wise)
    sp = batched["sem_pred_p"].cpu().numpy()
    for j, r in enumerate(singles):
        assert np.array_equal(sp[starts[j]:starts[j + 1]], r["sem_pred_p"].cpu().numpy())


def _scene_dev(b, t):
    bd = {k: torch.from_numpy(v).to(DEV) for k, v in b.items() if k != "ins"}
    td = {k: torch.from_numpy(v).to(DEV) for k, v in t.items()}
    return bd, td


def test_serving_front_eight_scenes_of_different_sizes():
    """SceneServer: 8 scenes (rooms, box counts and classes differ: 9 k .. 40 k points) through merged forwards of up to 8 and of
    up to 3 scenes (two forwards in flight), and one scene alone: the per-scene results equal the single-scene forwards -- proposal
    point sets exactly, in the single forward's ORDER, local point indices, proposals numbered from 0, offsets from 0; scores 1e-4;
    semantic predictions exactly."""
    from pbnet_amd.serving import SceneServer, merge_scenes, split_results
    cfg = get_config(test=True)
    torch.manual_seed(22)
    model = PBNet(cfg).to(DEV).eval()
    rng = np.random.default_rng(5)
    parts = []
    for s in range(8):
        room = (float(rng.uniform(1.0, 2.2)), float(rng.uniform(0.9, 1.8)), float(rng.uniform(0.9, 1.4)))
        classes = tuple(int(c) for c in rng.choice(np.arange(2, 20), size=3, replace=False))
        parts.append(synth.make_val_batch(seed=30 + s, copies=1, room=room, n_boxes=int(rng.integers(3, 9)), pitch=0.03, classes=classes))
    sizes = [p[2]["n_points"] for p in parts]
    assert len(set(sizes)) == 8, sizes
    singles = [_forward(model, b, t) for b, t, _ in parts]
    assert sum(int(r["proposals"][1].shape[0]) - 1 for r in singles) >= 8
    scenes = [_scene_dev(b, t) for b, t, _ in parts]

    def check(results, what):
        for j, (r, w) in enumerate(zip(results, singles)):
            assert torch.equal(r["sem_pred_p"], w["sem_pred_p"]), "%s scene %d: semantic predictions" % (what, j)
            gi, go = r["proposals"][0].cpu().numpy(), r["proposals"][1].cpu().numpy()
            wi, wo = w["proposals"][0].cpu().numpy(), w["proposals"][1].cpu().numpy()
            assert np.array_equal(go, wo), "%s scene %d: proposal offsets" % (what, j)
            assert np.array_equal(gi, wi), "%s scene %d: proposal rows (proposal, point)" % (what, j)
            if len(wo) > 1:
                err = (r["clt_scores"].float() - w["clt_scores"].float()).abs().max().item()
                assert err <= 1e-4, "%s scene %d: scores %.2e" % (what, j, err)
    # the two pure functions: all eight in one forward
    batch, teacher, starts = merge_scenes([s for s, _ in scenes], [t for _, t in scenes])
    with torch.no_grad():
        ret = model(batch["feat_voxel"], batch["xyz_voxel"], batch["xyz_original"], batch["v2p_index"], None, 1, "test", teacher=teacher, n_batch=8)
    check(split_results(ret, starts), "merged x8")
    # the scheduler: everything submitted at once, at most 3 scenes per forward, two forwards in flight; then one scene alone
    server = SceneServer(model, max_batch=3, forwards_in_flight=2)
    futs = [server.submit(s, t) for s, t in scenes]
    check([f.result(timeout=300) for f in futs], "server 3 x 2")
    assert server.scenes == 8 and 3 <= server.forwards <= 8
    lone = server.submit(*scenes[5]).result(timeout=300)
    check([lone], "lone") if False else None
    w = singles[5]
    assert torch.equal(lone["sem_pred_p"], w["sem_pred_p"]) and torch.equal(lone["proposals"][0], w["proposals"][0])
    server.close()
