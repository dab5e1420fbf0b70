"""CPU tier: the two pure functions of the serving front (pbnet_amd/serving.py) on CPU tensors -- no GPU, no model.
merge_scenes: batch column = scene number, v2p_index shifted by the voxels in front, point_starts.  split_results: a merged
forward's result (proposals of the scenes INTERLEAVED, as PBNet.forward orders them class-major) comes back per scene in the
reference's form -- proposals numbered from 0 in their original relative order, point indices local, offsets from 0, scores
aligned; scenes without proposals get empty results; a single scene passes through."""
import numpy as np
import torch

from pbnet_amd.serving import merge_scenes, split_results


def _scene(rng, n_vox, n_pts):
    xyz_voxel = torch.from_numpy(rng.integers(0, 50, (n_vox, 4)).astype(np.int32))
    xyz_voxel[:, 0] = 7          # whatever the caller left in the batch column
    return {"xyz_voxel": xyz_voxel, "feat_voxel": torch.randn(n_vox, 6), "xyz_original": torch.randn(n_pts, 3),
            "v2p_index": torch.from_numpy(rng.integers(0, n_vox, n_pts).astype(np.int64))}


def test_merge_scenes_batch_column_offsets_and_starts():
    rng = np.random.default_rng(0)
    scenes = [_scene(rng, 40, 100), _scene(rng, 25, 60), _scene(rng, 33, 81)]
    teachers = [{"sem_score": torch.randn(s["xyz_original"].shape[0], 20), "offset": torch.randn(s["xyz_original"].shape[0], 3)} for s in scenes]
    batch, teacher, starts = merge_scenes(scenes, teachers)
    assert starts == [0, 100, 160, 241]
    assert batch["xyz_voxel"].shape == (98, 4) and batch["v2p_index"].shape == (241,)
    v0 = 0
    for j, s in enumerate(scenes):
        nv, (p0, p1) = s["xyz_voxel"].shape[0], (starts[j], starts[j + 1])
        assert bool((batch["xyz_voxel"][v0:v0 + nv, 0] == j).all())
        assert torch.equal(batch["xyz_voxel"][v0:v0 + nv, 1:], s["xyz_voxel"][:, 1:])
        assert torch.equal(batch["v2p_index"][p0:p1], s["v2p_index"] + v0)
        assert torch.equal(batch["feat_voxel"][v0:v0 + nv], s["feat_voxel"]) and torch.equal(batch["xyz_original"][p0:p1], s["xyz_original"])
        assert torch.equal(teacher["offset"][p0:p1], teachers[j]["offset"])
        assert int(s["xyz_voxel"][0, 0]) == 7, "the caller's tensors are not modified"
        v0 += nv
    lone, t1, st1 = merge_scenes(scenes[:1], None)
    assert st1 == [0, 100] and t1 is None and bool((lone["xyz_voxel"][:, 0] == 0).all()) and lone["v2p_index"] is scenes[0]["v2p_index"]


def test_split_results_returns_every_scene_its_own_proposals():
    rng = np.random.default_rng(1)
    n_pts = [120, 0 + 75, 200, 90]                      # scene 1 will have no proposal at all
    starts = [0] + list(np.cumsum(n_pts))
    per_scene = []
    for j, n in enumerate(n_pts):
        props = []
        for _ in range(0 if j == 1 else int(rng.integers(2, 6))):
            size = int(rng.integers(3, 30))
            props.append((np.sort(rng.choice(n, size=size, replace=False)), float(rng.random())))
        per_scene.append(props)
    # the merged forward's order: interleave the scenes' proposals (class-major in the real forward), each scene's own order kept
    order = []
    cursors = [0] * len(n_pts)
    while any(cursors[j] < len(per_scene[j]) for j in range(len(n_pts))):
        j = int(rng.integers(0, len(n_pts)))
        if cursors[j] < len(per_scene[j]):
            order.append((j, cursors[j]))
            cursors[j] += 1
    rows, off, scores = [], [0], []
    for p, (j, q) in enumerate(order):
        pts, sc = per_scene[j][q]
        rows.append(np.stack([np.full(len(pts), p), pts + starts[j]], 1))
        off.append(off[-1] + len(pts))
        scores.append(sc)
    sem = torch.from_numpy(rng.integers(0, 20, int(starts[-1])))
    ret = {"sem_pred_p": sem, "proposals": (torch.from_numpy(np.concatenate(rows)).long(), torch.tensor(off, dtype=torch.int64), None, None),
           "clt_scores": torch.tensor(scores, dtype=torch.float32)}
    out = split_results(ret, [int(s) for s in starts])
    assert len(out) == 4
    for j, res in enumerate(out):
        assert torch.equal(res["sem_pred_p"], sem[starts[j]:starts[j + 1]])
        idx, o = res["proposals"][0].numpy(), res["proposals"][1].numpy()
        assert o[0] == 0 and len(o) == len(per_scene[j]) + 1 and res["clt_scores"].shape[0] == len(per_scene[j])
        for q, (pts, sc) in enumerate(per_scene[j]):
            seg = idx[o[q]:o[q + 1]]
            assert np.array_equal(seg[:, 0], np.full(len(pts), q)) and np.array_equal(seg[:, 1], pts)
            assert abs(float(res["clt_scores"][q]) - sc) < 1e-7
    # no proposals at all; a single scene passes through untouched
    empty = {"sem_pred_p": sem, "proposals": (torch.zeros(0, 2, dtype=torch.int64), torch.zeros(1, dtype=torch.int64), None, None),
             "clt_scores": torch.zeros(0)}
    for res in split_results(empty, [int(s) for s in starts]):
        assert res["proposals"][0].shape == (0, 2) and res["proposals"][1].tolist() == [0] and res["clt_scores"].numel() == 0
    one = split_results({"sem_pred_p": sem[:120], "proposals": ret["proposals"], "clt_scores": ret["clt_scores"]}, [0, 120])
    assert one[0]["proposals"][0] is ret["proposals"][0] and one[0]["clt_scores"] is ret["clt_scores"]
