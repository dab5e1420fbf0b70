"""GPU tier: the HIP grouping pipeline, called through the C ABI, against the golden fixtures, the CPU oracle on
fresh seeded inputs, and size-independent properties at BASELINE.json's full scene size.

Bar: bit-exact for every integer output AND for the fp32 centre bits (the arithmetic contract fixes the order)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import pb_cluster_ref as oracle
from pbnet_amd import pbnet_ops, synth

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cluster_*.npz")))
DEV = "cuda:0"


def _run(off, org, sem, seg, radius, min_pts, nv=True, general=None):
    sem = np.asarray(sem)
    if general is None:
        start = np.concatenate([[0], np.cumsum(seg)])
        general = any(len(np.unique(sem[a:b])) > 1 for a, b in zip(start[:-1], start[1:]))
    res = pbnet_ops.cluster_device(torch.from_numpy(np.ascontiguousarray(off, np.float32)).to(DEV),
                                   torch.from_numpy(np.ascontiguousarray(org, np.float32)).to(DEV),
                                   torch.from_numpy(np.ascontiguousarray(sem, np.int32)).to(DEV),
                                   torch.from_numpy(np.ascontiguousarray(seg, np.int32)).to(DEV), radius, min_pts,
                                   nv_flag=nv, general_sem=general)
    c = int(res.n_clusters.item())
    assert c >= 0
    out = dict(cluster_id=res.cluster_id.cpu().numpy(), cluster_num=res.cluster_num.cpu().numpy(),
               den_queue=res.den.cpu().numpy(), center=res.centers[:3 * c].cpu().numpy().reshape(c, 3),
               clt_sem=res.clt_sem[:c].cpu().numpy(), member_start=res.member_start[:c + 1].cpu().numpy(),
               member_idx=res.member_idx.cpu().numpy())
    return out


def _assert_same(got, want):
    for k in ("cluster_id", "cluster_num", "den_queue", "clt_sem"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["center"].view(np.int32), want["center"].view(np.int32)), "centre bits"
    # members CSR == torch.nonzero(cluster_id == c) per cluster (PBNet.py:204)
    c = want["center"].shape[0]
    ms = got["member_start"]
    for k in range(c):
        assert np.array_equal(got["member_idx"][ms[k]:ms[k + 1]], np.nonzero(want["cluster_id"] == k)[0])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[8:-4] for p in GOLDEN])
def test_golden(path):
    g = dict(np.load(path))
    got = _run(g["off"], g["org"], g["sem"], g["seg"], float(g["radius"]), int(g["min_pts"]), nv=bool(g["nv_flag"]))
    _assert_same(got, g)


@pytest.mark.parametrize("path", [p for p in GOLDEN if "G8" not in p][:6])
def test_general_path_equals_fast_path_on_uniform_classes(path):
    g = dict(np.load(path))
    got = _run(g["off"], g["org"], g["sem"], g["seg"], float(g["radius"]), int(g["min_pts"]), nv=bool(g["nv_flag"]),
               general=True)
    _assert_same(got, g)


def test_reference_shaped_ops(golden_dir):
    """pbnet_ops.cluster (pbnet_ops.py:82) and PB_lib.binary_cluster (PB_lib_api.cpp:7) with CPU tensors, as the
    reference's own PBNet.forward calls them (PBNet.py:176)."""
    from pbnet_amd import PB_lib
    g = dict(np.load(os.path.join(golden_dir, "cluster_G6.npz")))
    off, org = torch.from_numpy(g["off"]), torch.from_numpy(g["org"])
    sem, seg = torch.from_numpy(g["sem"]).long(), torch.from_numpy(g["seg"])
    cid, cnum, den, ctr = pbnet_ops.cluster(off, org, sem, seg, 0.04, 31, len(seg))
    assert cid.device.type == "cpu" and cid.dtype == torch.int32
    assert np.array_equal(cid.numpy(), g["cluster_id"]) and np.array_equal(cnum.numpy(), g["cluster_num"])
    assert np.array_equal(den.numpy(), g["den_queue"] + 1)
    assert np.array_equal(ctr.numpy().view(np.int32), g["center"].reshape(-1).view(np.int32))
    # native-module form: outputs in place, center/clt_sem resized
    n = off.shape[0]
    x, y, z = (off[:, k].contiguous() for k in range(3))
    xo, yo, zo = (org[:, k].contiguous() for k in range(3))
    cluster_id = torch.full((n,), -1, dtype=torch.int32)
    cluster_num = torch.zeros(len(seg), dtype=torch.int32)
    den_q = torch.zeros(n, dtype=torch.int32)
    center = torch.zeros(n, dtype=torch.float32)
    clt_sem = torch.zeros(n, dtype=torch.int32)
    l1 = x.abs() + y.abs() + z.abs()
    mapper = torch.cat([torch.arange(int(k)) for k in seg]).int()
    PB_lib.binary_cluster(x, y, z, l1, mapper, xo, yo, zo, sem.int(), seg, torch.ones(18) * 0.04,
                          (torch.ones(18) * 31).int(), cluster_id, cluster_num, den_q, center, clt_sem, len(seg), 0.05,
                          True)
    assert np.array_equal(cluster_id.numpy(), g["cluster_id"]) and np.array_equal(den_q.numpy(), g["den_queue"])
    assert center.shape[0] == 3 * g["center"].shape[0] and clt_sem.shape[0] == g["clt_sem"].shape[0]
    assert np.array_equal(clt_sem.numpy(), g["clt_sem"])


def test_empty_and_degenerate_inputs():
    z3 = np.zeros((0, 3), np.float32)
    got = _run(z3, z3, np.zeros(0, np.int32), np.array([0, 0], np.int32), 0.04, 31)
    assert got["cluster_num"].tolist() == [0, 0] and got["cluster_id"].shape == (0,)
    # a single point; all points identical
    one = np.array([[1.0, 2.0, 3.0]], np.float32)
    got = _run(one, one, np.array([17]), np.array([1]), 0.04, 31)
    assert got["cluster_id"].tolist() == [-1] and got["den_queue"].tolist() == [0]
    same = np.tile(one, (100, 1))
    got = _run(same, same, np.full(100, 17), np.array([100]), 0.04, 31)
    want = oracle.binary_cluster(same, same, np.full(100, 17), np.array([100]), 0.04, 31)
    assert np.array_equal(got["cluster_id"], want["cluster_id"]) and got["den_queue"].tolist() == [99] * 100
    # invalid: lengths do not add up -> n_clusters = -1
    res = pbnet_ops.cluster_device(torch.zeros(5, 3, device=DEV), torch.zeros(5, 3, device=DEV),
                                   torch.full((5,), 17, dtype=torch.int32, device=DEV),
                                   torch.tensor([2, 2], dtype=torch.int32, device=DEV), 0.04, 31)
    assert int(res.n_clusters.item()) == -1


def test_far_away_and_negative_coordinates():
    rng = np.random.default_rng(5)
    P = np.concatenate([rng.normal(0, 0.01, (80, 3)) + [-1500.0, 2000.0, -3.0], rng.normal(0, 0.01, (90, 3)) + [1e4, -1e4, 1e4],
                        rng.normal(0, 0.01, (70, 3))]).astype(np.float32)
    sem = np.full(len(P), 17)
    want = oracle.binary_cluster(P, P, sem, [len(P)], 0.04, 31)
    got = _run(P, P, sem, np.array([len(P)]), 0.04, 31)
    _assert_same(got, want)


def _scene_groups(seed, pitch, room, n_boxes, copies=3):
    """All (class, batch copy) groups of a teacher-forced synthetic scene, laid out as PBNet.forward lays them out
    (PBNet.py:151-173): classes ascending, inside a class the batch copies in order, inside a copy ascending index."""
    sc = synth.synth_room(seed=seed, pitch=pitch, room=room, n_boxes=n_boxes)
    sem_pred, offset = synth.teacher_forced_heads(sc, seed=seed)
    offs, orgs, sems, segs = [], [], [], []
    for cls in range(2, 20):
        idx = np.nonzero(sem_pred == cls)[0]
        if len(idx) * copies < 3917 * 0.0:  # no class-size gate here: keep tiny groups too (edge cases)
            continue
        for c in range(copies):
            th = np.deg2rad([63.0, 183.0, 303.0][c % 3])
            R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
            orgs.append((sc["xyz"][idx] @ R.T).astype(np.float32))
            offs.append(((sc["xyz"][idx] + offset[idx]) @ R.T).astype(np.float32))
            sems.append(np.full(len(idx), cls, np.int32))
            segs.append(len(idx))
    return np.concatenate(offs), np.concatenate(orgs), np.concatenate(sems), np.array(segs, np.int32)


def test_all_class_groups_in_one_launch_vs_oracle():
    """The MI355X-native call shape: every (class, copy) group of a scene in ONE launch (54 segments)."""
    off, org, sem, seg = _scene_groups(seed=7, pitch=0.04, room=(3.0, 2.4, 2.0), n_boxes=20)
    want = oracle.binary_cluster(off, org, sem, seg, 0.04, 31)
    got = _run(off, org, sem, seg, 0.04, 31, general=False)
    _assert_same(got, want)
    assert want["cluster_num"].sum() > 20


def test_full_size_scene_properties_and_oracle():
    """BASELINE configs[1] size (~175k points, 3 TTA copies -> ~300k grouped points): bit-exact against the oracle and
    size-independent properties (permutation of segments, idempotent relabel, den symmetry)."""
    off, org, sem, seg = _scene_groups(seed=2, pitch=0.0225, room=(4.0, 3.2, 2.6), n_boxes=12, copies=3)
    got = _run(off, org, sem, seg, 0.04, 31, general=False)
    want = oracle.binary_cluster(off, org, sem, seg, 0.04, 31)
    _assert_same(got, want)
    n = len(off)
    assert n > 150000
    # every point of a non-empty class group ends assigned when the group has a surviving cluster
    start = np.concatenate([[0], np.cumsum(seg)])
    for b in range(len(seg)):
        ids = got["cluster_id"][start[b]:start[b + 1]]
        if got["cluster_num"][b] > 0:
            assert (ids >= 0).all()
        else:
            assert (ids == -1).all()
    # ids are contiguous and ordered by first (seed) occurrence within the kept set
    assert got["cluster_id"].max() + 1 == got["cluster_num"].sum()
    # segment order permutation: reversing the segment order permutes ids but not the partition / den / sizes
    order = np.arange(len(seg))[::-1]
    perm = np.concatenate([np.arange(start[b], start[b + 1]) for b in order])
    got2 = _run(off[perm], org[perm], sem[perm], seg[order], 0.04, 31, general=False)
    assert np.array_equal(got2["den_queue"], got["den_queue"][perm])
    a, b2 = got["cluster_id"][perm], got2["cluster_id"]
    pairs = np.unique(np.stack([a, b2], 1), axis=0)
    assert len(pairs) == len(np.unique(a)) == len(np.unique(b2))  # one-to-one relabelling


def test_get_iou_matches_oracle():
    rng = np.random.default_rng(3)
    n, n_inst, n_prop = 20000, 37, 64
    labels = rng.integers(-1, n_inst, n).astype(np.int64)
    labels[labels < 0] = -100
    pointnum = np.array([(labels == i).sum() for i in range(n_inst)], np.int32)
    lens = rng.integers(0, 900, n_prop)
    lens[5] = 0
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, n, offs[-1]).astype(np.int32)
    want = oracle.get_iou(idx, offs, labels, pointnum)
    t = lambda a: torch.from_numpy(a).to(DEV)
    got = pbnet_ops.get_iou(t(idx), t(offs), t(labels), t(pointnum)).cpu().numpy()
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
    iou2, mask_label = pbnet_ops.cal_iou_and_masklabel(t(idx), t(offs), t(labels), t(pointnum),
                                                       torch.rand(int(offs[-1]), 1, device=DEV), 0)
    assert np.array_equal(iou2.cpu().numpy().view(np.int32), want.view(np.int32))
    assert set(np.unique(mask_label.cpu().numpy())) <= {-1.0, 0.0, 1.0}


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[8:-4] for p in GOLDEN])
def test_capacity_mode_equals_exact_mode(path):
    """pbn_binary_cluster with n_points as a CAPACITY (flags bit 1: the number of points that exist is sum(seg_len), a
    device-side count -- the capacity-planned forward never reads it back): rows beyond the count hold garbage and must
    neither be read nor change any output of the rows that exist."""
    g = dict(np.load(path))
    n = g["off"].shape[0]
    pad = n // 3 + 100
    rng = np.random.default_rng(1)
    junk3 = rng.normal(0, 50, (pad, 3)).astype(np.float32)
    junk3[::7] = np.nan
    off = np.concatenate([g["off"], junk3]).astype(np.float32)
    org = np.concatenate([g["org"], junk3[::-1]]).astype(np.float32)
    sem = np.concatenate([g["sem"], np.full(pad, 77, np.int32)]).astype(np.int32)      # an invalid class id: must never be looked at
    start = np.concatenate([[0], np.cumsum(g["seg"])])
    general = any(len(np.unique(np.asarray(g["sem"])[a:b])) > 1 for a, b in zip(start[:-1], start[1:]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    res = pbnet_ops.cluster_device(t(off), t(org), t(sem), t(g["seg"].astype(np.int32)), float(g["radius"]), int(g["min_pts"]),
                                   nv_flag=bool(g["nv_flag"]), general_sem=general, capacity=True)
    c = int(res.n_clusters.item())
    assert c == g["center"].shape[0]
    got = dict(cluster_id=res.cluster_id[:n].cpu().numpy(), cluster_num=res.cluster_num.cpu().numpy(),
               den_queue=res.den[:n].cpu().numpy(), center=res.centers[:3 * c].cpu().numpy().reshape(c, 3),
               clt_sem=res.clt_sem[:c].cpu().numpy(), member_start=res.member_start[:c + 1].cpu().numpy(),
               member_idx=res.member_idx.cpu().numpy())
    _assert_same(got, g)
