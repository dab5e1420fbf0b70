"""GPU tier: the fused stage-glue launches (csrc/stages.hip) against the tensor-op form they replace
(/root/reference/network/PBNet.py:182-247 and friends).  Integer outputs bit-exact; feature rows bit-exact (copies)."""
import numpy as np
import pytest
import torch

from pbnet_amd import stage_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _entries(rng, n_clusters, sizes, n_scenes):
    """Random local-scene plan: each scene = 1..4 entries (clusters, possibly repeated across scenes)."""
    ent_cluster, ent_weight, scene_len = [], [], []
    for _ in range(n_scenes):
        k = int(rng.integers(1, 5))
        scene_len.append(k)
        for j in range(k):
            ent_cluster.append(int(rng.integers(0, n_clusters)))
            ent_weight.append(1.0 if j == 0 else float(rng.uniform(0.05, 0.5)))
    return ent_cluster, ent_weight, scene_len


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_local_scene_rows_matches_tensor_ops(dtype):
    rng = np.random.default_rng(3)
    n_points, m, n_clusters, c, n_cls = 5000, 3000, 37, 32, 20
    sizes = rng.integers(1, 120, n_clusters)
    sizes[5] = 700                                              # one long cluster
    member_start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    total = int(member_start[-1])
    member_idx = rng.integers(0, m, total).astype(np.int32)
    ins_ind = torch.from_numpy(rng.permutation(n_points)[:m].astype(np.int64)).to(DEV)
    xyz = torch.from_numpy(rng.uniform(-3, 6, (n_points, 3)).astype(np.float32)).to(DEV)
    point_feat = torch.randn(n_points, c, device=DEV).to(dtype)
    sem_score = torch.softmax(torch.randn(n_points, n_cls, device=DEV), 1).to(dtype)
    sem_pred = sem_score.float().max(1)[1]
    ent_cluster, ent_weight, scene_len = _entries(rng, n_clusters, sizes, 23)
    ec = torch.tensor(ent_cluster, dtype=torch.long)
    ent_rows = torch.from_numpy(sizes.astype(np.int64))[ec]
    n_ent = len(ent_cluster)
    row_start = torch.zeros(n_ent + 1, dtype=torch.int32)
    row_start[1:] = torch.cumsum(ent_rows, 0)
    n_rows = int(row_start[-1])
    ent_scene = torch.repeat_interleave(torch.arange(len(scene_len), dtype=torch.int32), torch.tensor(scene_len))
    ms = torch.from_numpy(member_start)
    packed = torch.cat([row_start, ms[:-1][ec].to(torch.int32), ent_scene,
                        torch.tensor(ent_weight, dtype=torch.float32).view(torch.int32)]).to(DEV)
    midx = torch.from_numpy(member_idx).to(DEV)
    voxel = 0.02
    point_idx, row_scene, coords, feat = stage_ops.local_scene_rows(packed, n_ent, n_rows, midx, ins_ind, xyz, voxel,
                                                                   point_feat, sem_score, sem_pred)
    # tensor-op form (what PBNet.forward does with autograd enabled)
    row_ent = torch.repeat_interleave(torch.arange(n_ent, device=DEV), ent_rows.to(DEV))
    ent_first = (torch.cumsum(ent_rows, 0) - ent_rows).to(DEV)
    pos = torch.arange(n_rows, device=DEV) - ent_first[row_ent]
    want_p = ins_ind[midx[ms[:-1][ec].to(DEV)[row_ent] + pos].long()]
    want_scene = ent_scene.to(DEV).long()[row_ent]
    want_w = torch.tensor(ent_weight, dtype=torch.float32, device=DEV)[row_ent]
    want_feat = torch.cat([point_feat[want_p], sem_score[want_p, sem_pred[want_p]].view(-1, 1),
                           want_w.view(-1, 1).to(dtype)], 1)
    want_coords = torch.cat([want_scene.view(-1, 1).to(torch.int32), torch.floor(xyz[want_p] / voxel).to(torch.int32)], 1)
    assert torch.equal(point_idx, want_p)
    assert torch.equal(row_scene, want_scene)
    assert torch.equal(coords, want_coords)
    assert torch.equal(feat.view(torch.uint8), want_feat.contiguous().view(torch.uint8))


def test_gather_pad_rows():
    from pbnet_amd import _native as N
    import ctypes
    x = torch.randn(1000, 34, device=DEV).to(torch.bfloat16)
    idx = torch.randperm(1000, device=DEV)[:777]
    out = torch.empty(777, 40, dtype=torch.bfloat16, device=DEV)
    rc = N.lib().pbn_gather_pad_rows(ctypes.c_void_p(x.data_ptr()), 68, 68, ctypes.c_void_p(idx.data_ptr()), 777,
                                     ctypes.c_void_p(out.data_ptr()), 80, N.current_stream())
    assert rc == 0
    assert torch.equal(out[:, :34], x[idx]) and (out[:, 34:] == 0).all()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("n_out,sigmoid,hidden", [(20, False, 16), (3, False, 16), (1, True, 16), (32, False, 32)])
def test_mlp_rows_matches_modules(dtype, tol, n_out, sigmoid, hidden):
    """PBNet.py:43-82 heads: Linear -> BatchNorm(eval) -> PReLU -> Linear [-> Sigmoid], with a two-level row index."""
    import pbnet_amd.MinkowskiEngine as ME
    torch.manual_seed(n_out)
    layers = [ME.MinkowskiLinear(32, hidden, bias=False), ME.MinkowskiBatchNorm(hidden), ME.MinkowskiPReLU(),
              ME.MinkowskiLinear(hidden, n_out, bias=True)]
    if sigmoid:
        layers.append(ME.MinkowskiSigmoid())
    head = torch.nn.Sequential(*layers).eval()
    with torch.no_grad():
        head[1].bn.running_mean.normal_(0, 0.3)
        head[1].bn.running_var.uniform_(0.5, 2.0)
        head[1].bn.weight.uniform_(0.5, 1.5)
        head[1].bn.bias.normal_(0, 0.2)
        head[2].module.weight.fill_(0.2)
    x = torch.randn(3000, 32)
    idx_a = torch.randint(0, 2500, (4000,))
    idx_b = torch.randperm(3000)[:2500]
    with torch.no_grad():
        h = head[0].linear(x)
        h = head[2].module(head[1].bn(h))
        want = head[3].linear(h)
        if sigmoid:
            want = torch.sigmoid(want)
        want = want[idx_b[idx_a]]
    head = head.to(DEV)
    got = stage_ops.mlp_rows(head, x.to(DEV).to(dtype), idx_a.to(DEV), idx_b.to(DEV))
    assert got.shape == (4000, n_out) and got.dtype == dtype
    err = (got.float().cpu() - want).abs().max().item()
    # fp32: absolute; bf16: bound relative to the output range (not the parity configuration)
    assert err <= (tol if dtype == torch.float32 else tol * max(1.0, want.abs().max().item())), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sem_argmax_table_and_select_points(dtype):
    """PBNet.py:134,151-170: arg-max / own-class softmax / population table, then the stable class-major selection."""
    torch.manual_seed(4)
    n, s, nb = 70001, 20, 3
    score = (torch.randn(n, s) * 3).to(dtype).to(DEV)
    score[:5000, 7] += 20                                           # long single-class runs
    batch = torch.randint(0, nb, (n,), dtype=torch.int32, device=DEV)
    sem_pred, sem_prob, table, block_hist = stage_ops.sem_argmax_table(score, batch, nb)
    sf = score.float()
    want_pred = sf.max(1)[1]
    assert torch.equal(sem_pred, want_pred)
    want_prob = torch.softmax(sf, 1).gather(1, want_pred.view(-1, 1)).view(-1)
    assert (sem_prob.float() - want_prob).abs().max().item() <= (1e-6 if dtype == torch.float32 else 4e-3)
    want_table = torch.bincount(want_pred * nb + batch.long(), minlength=s * nb).view(s, nb)
    assert torch.equal(table.long(), want_table)
    assert torch.equal(block_hist.sum(0).long(), want_table.sum(1))
    # selection: drop classes 0, 1 and two more
    per_class = want_table.sum(1).cpu()
    classes = [c for c in range(2, s) if c not in (5, 11)]
    class_base = torch.full((s,), -1, dtype=torch.int32)
    run = 0
    for c in classes:
        class_base[c] = run
        run += int(per_class[c])
    xyz = torch.randn(n, 3, device=DEV)
    offset = torch.randn(n, 3, device=DEV).to(dtype)
    ins_ind, ins_orig, ins_off, ins_sem = stage_ops.select_points(sem_pred, class_base.to(DEV), block_hist, xyz, offset, run)
    keep = torch.zeros(s, dtype=torch.bool)
    keep[classes] = True
    key = torch.where(keep.to(DEV)[want_pred], want_pred, torch.full_like(want_pred, s))
    order = torch.sort(key, stable=True)[1][:run]
    assert torch.equal(ins_ind, order)
    assert torch.equal(ins_orig, xyz[order])
    assert torch.equal(ins_off, xyz[order] + offset[order].float())
    assert torch.equal(ins_sem, want_pred[order].to(torch.int32))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_proposal_compaction_matches_get_proposal(dtype):
    """PBNet.py:317-347 (+240-252): threshold, per-scene counts, dense renumbering of surviving scenes, order kept."""
    torch.manual_seed(8)
    n_scenes, n_points = 41, 9000
    lens = torch.randint(0, 900, (n_scenes,))
    lens[3] = 0
    lens[17] = 5000
    row_scene = torch.repeat_interleave(torch.arange(n_scenes), lens).to(DEV)
    r = int(row_scene.shape[0])
    point_idx = torch.randint(0, n_points, (r,), device=DEV)
    score = torch.rand(r, 1, device=DEV).to(dtype)
    score[row_scene == 9] = 0.1                                     # a scene that dies entirely
    xyz = (torch.rand(n_points, 3, device=DEV) * 8 - 2)
    feat = torch.randn(n_points, 32, device=DEV).to(dtype)
    thd, scale, voxel = 0.45, 1, 0.02
    per_scene_d, block_cnt = stage_ops.mask_count(score, thd, row_scene, n_scenes)
    keep = score.view(-1).float() > thd
    want_per_scene = torch.bincount(row_scene[keep], minlength=n_scenes)
    assert torch.equal(per_scene_d.long(), want_per_scene)
    per_scene = want_per_scene.cpu()
    alive = per_scene > 0
    dense_of = (torch.cumsum(alive.to(torch.int32), 0) - 1).to(torch.int32)
    total = int(per_scene.sum())
    pidx, pms, coords, f3 = stage_ops.proposal_rows(score, thd, row_scene, point_idx, dense_of.to(DEV), block_cnt, total,
                                                    xyz, scale, voxel, feat)
    valid = torch.nonzero(keep).view(-1)
    want_idx = torch.stack([dense_of.to(DEV).long()[row_scene[valid]], point_idx[valid]], 1)
    assert torch.equal(pidx, want_idx)
    assert torch.equal(pms, score[valid].view(-1))
    p = want_idx[:, 1]
    want_c = torch.cat([want_idx[:, 0:1].to(torch.int32), torch.floor(xyz[p] * scale / voxel).to(torch.int32)], 1)
    assert torch.equal(coords, want_c)
    assert torch.equal(f3, feat[p])
