"""GPU tier, path level (SURVEY.md 8c fixture P1): PBNet.forward on the MI355X path against the CPU restatement
(oracle/pbnet_ref.py).  Stage 1 (backbone + heads) is compared at 1e-4; stage 2 (grouping, local scenes, mask
branch, proposals, score branch) is driven by IDENTICAL stage-1 tensors on both sides, so every integer output must
match exactly and the scores within 1e-4."""
import numpy as np
import pytest
import torch

from oracle import pbnet_ref
from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet, model_fn_eval

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def make_batch(seed=1, copies=3):
    """3 rotated copies of a 6-box scene whose boxes alternate between two small-threshold classes."""
    batch, teacher, _ = synth.make_val_batch(seed=seed, copies=copies, room=(1.6, 1.3, 1.2), n_boxes=6, pitch=0.03,
                                             classes=(17, 10))
    t = torch.from_numpy
    return {k: t(v) for k, v in batch.items()}, {k: t(v) for k, v in teacher.items()}


@pytest.fixture(scope="module")
def setup():
    cfg = get_config(test=True)
    torch.manual_seed(22)
    model = PBNet(cfg)
    g = torch.Generator().manual_seed(5)
    for mod in model.modules():  # non-trivial BN statistics so eval-mode folding is exercised
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    batch, teacher = make_batch()
    return cfg, model, sd, batch, teacher


def test_stage1_backbone_and_heads(setup):
    cfg, model, sd, batch, _ = setup
    with torch.no_grad():
        got = model.backbone_stage(batch["feat_voxel"].to(DEV), batch["xyz_voxel"].to(DEV), batch["v2p_index"].to(DEV))
    want = pbnet_ref.backbone_stage(sd, batch["feat_voxel"], batch["xyz_voxel"], batch["v2p_index"])
    for k in ("point_feat_p", "sem_pred_score_p", "sem_pred_score_sfp", "offset_pred_p"):
        w = want[k]
        err = (got[k].cpu() - w).abs().max().item()
        print("%s: max |diff| %.3e (abs tol %.0e)" % (k, err, TOL))
        assert err <= TOL, (k, err)            # ABSOLUTE 1e-4 (BASELINE.json north_star)
    assert torch.equal(got["batch_head_p"].cpu().long(), want["batch_head_p"].long())


def _stage2_inputs(sd, batch, teacher):
    s1 = pbnet_ref.backbone_stage(sd, batch["feat_voxel"], batch["xyz_voxel"], batch["v2p_index"])
    s1["sem_pred_score_p"] = teacher["sem_score"]
    s1["sem_pred_score_sfp"] = torch.softmax(teacher["sem_score"], 1)
    s1["offset_pred_p"] = teacher["offset"]
    s1["sem_pred_p"] = s1["sem_pred_score_p"].max(1)[1]
    return s1


@pytest.mark.parametrize("task", ["test", "eval"])
def test_stage2_grouping_to_scores(setup, task):
    cfg, model, sd, batch, teacher = setup
    s1 = _stage2_inputs(sd, batch, teacher)
    ins = None if task == "test" else batch["ins"]
    want = pbnet_ref.cluster_stage(sd, cfg, s1, batch["xyz_original"], ins, task)
    assert want["n_local_scenes"] >= 12 and sum(int(v.sum()) for v in want["n_clusters_per_class"].values()) >= 12
    s1d = {k: v.to(DEV) for k, v in s1.items()}
    with torch.no_grad():
        got = model.cluster_stage(s1d, batch["xyz_original"].to(DEV), None if ins is None else ins.to(DEV), task)
    gi, go, gv, gm = got["proposals"]
    wi, wo, wv, wm = want["proposals"]
    assert gi.dtype == torch.int64 and go.dtype == torch.int64
    assert torch.equal(go.cpu(), wo), "proposals_offset"
    assert torch.equal(gi.cpu(), wi), "proposals_idx"
    assert torch.equal(gv.cpu(), wv.long()), "surviving local scene ids"
    assert (gm.cpu() - wm).abs().max().item() <= TOL
    assert got["clt_scores"].shape == want["clt_scores"].shape
    assert (got["clt_scores"].cpu() - want["clt_scores"]).abs().max().item() <= TOL
    if task != "test":
        assert torch.equal(got["mask_scores"][1].cpu(), want["mask_scores"][1])
        assert (got["mask_scores"][0].cpu() - want["mask_scores"][0]).abs().max().item() <= TOL


def test_forward_and_model_fn_eval(setup):
    """End to end through the reference-shaped entry points, incl. the teacher hook and dict keys (PBNet.py:138-141,
    251-252,279; model_fn_eval :446-460)."""
    cfg, model, sd, batch, teacher = setup
    with torch.no_grad():
        ret = model(batch["feat_voxel"], batch["xyz_voxel"], batch["xyz_original"], batch["v2p_index"], None, 1, "test",
                    teacher=teacher)
        pred = model_fn_eval(batch, model, 1, cfg, teacher=teacher)
    assert set(ret) == {"sem_pred_p", "sem_pred_score_p", "offset_pred_p", "proposals", "clt_scores"}
    assert set(pred) == {"sem", "proposals", "clt_scores"}
    idx, off, _, ms = ret["proposals"]
    assert off[-1].item() == idx.shape[0] == ms.shape[0] and ret["clt_scores"].shape[0] == off.shape[0] - 1
    assert torch.equal(pred["proposals"][0], idx)
    # without the hook a random-init network finds no dense cores: the stage must come back empty, not fail
    with torch.no_grad():
        ret0 = model(batch["feat_voxel"], batch["xyz_voxel"], batch["xyz_original"], batch["v2p_index"], None, 1, "test")
    assert ret0["proposals"][1].shape[0] >= 1
    # cluster stage switched off below cluster_epoch (PBNet.py:144)
    model.cluster_epoch = 128
    with torch.no_grad():
        r = model(batch["feat_voxel"], batch["xyz_voxel"], batch["xyz_original"], batch["v2p_index"], None, 1, "test")
    model.cluster_epoch = cfg.cluster_epoch
    assert "proposals" not in r


def test_forward_degenerate_scenes(setup):
    """The data-dependent early exits of PBNet.py:144-279 on the fused inference path: no class passes the population
    gate; every mask score falls under the threshold (no proposal survives); both must return well-formed empties."""
    cfg, model, sd, batch, teacher = setup
    args = (batch["feat_voxel"], batch["xyz_voxel"], batch["xyz_original"], batch["v2p_index"], None, 1, "test")
    # (a) every point predicted as floor (class 0): nothing to group
    t0 = {"sem_score": torch.full_like(teacher["sem_score"], -5.0), "offset": teacher["offset"]}
    t0["sem_score"][:, 0] = 5.0
    with torch.no_grad():
        ret = model(*args, teacher=t0)
    assert ret["proposals"][0].shape == (0, 2) and ret["proposals"][1].shape[0] == 1 and ret["clt_scores"].numel() == 0
    assert torch.equal(ret["sem_pred_p"].cpu(), torch.zeros(batch["xyz_original"].shape[0], dtype=torch.int64))
    # (b) the mask head answers 0 everywhere: every local scene dies at the threshold
    bias = model.linear_binary[3].linear.bias
    keep = bias.detach().clone()
    try:
        with torch.no_grad():
            bias.fill_(-60.0)
            ret = model(*args, teacher=teacher)
        assert ret["proposals"][0].shape[0] == 0 and ret["proposals"][1].tolist() == [0] and ret["clt_scores"].numel() == 0
    finally:
        with torch.no_grad():
            bias.copy_(keep)
    with torch.no_grad():                                      # and the model is intact afterwards
        ret = model(*args, teacher=teacher)
    assert ret["proposals"][1].shape[0] > 1


def test_scenes_in_flight(setup):
    """Several scenes in flight on one GPU (one host thread + HIP stream each, what `bench.py --inflight` and a serving
    loop do): every result must be bit-identical to the one-at-a-time result -- no scratch buffer, cache or table may be
    shared between concurrent forwards."""
    import threading
    cfg, model, sd, _, _ = setup
    dev = torch.device(DEV)
    scenes = []
    for seed, boxes in ((1, 6), (2, 4), (3, 7)):
        b, t, _ = synth.make_val_batch(seed=seed, copies=3, room=(1.6, 1.3, 1.2), n_boxes=boxes, pitch=0.03, classes=(17, 10))
        scenes.append(({k: torch.from_numpy(v).to(dev) for k, v in b.items()}, {k: torch.from_numpy(v).to(dev) for k, v in t.items()}))

    def run(i):
        b, t = scenes[i]
        with torch.no_grad():
            r = model(b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"], None, 1, "test", teacher=t)
        return [r["proposals"][0].cpu(), r["proposals"][1].cpu(), r["proposals"][3].cpu(), r["clt_scores"].cpu(),
                r["sem_pred_p"].cpu(), r["offset_pred_p"].cpu()]

    want = [run(i) for i in range(len(scenes))]
    assert all(w[1].shape[0] > 1 for w in want)                # every scene yields proposals
    torch.cuda.synchronize()
    n_workers, rounds = 3, 4
    streams = [torch.cuda.Stream(dev) for _ in range(n_workers)]
    got, errors = [[] for _ in range(n_workers)], []

    def worker(w):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[w]):
                for r in range(rounds):
                    for i in range(len(scenes)):
                        j = (i + w + r) % len(scenes)
                        got[w].append((j, run(j)))
        except BaseException as e:
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(n_workers)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for w in range(n_workers):
        assert len(got[w]) == rounds * len(scenes)
        for j, res in got[w]:
            for a, b_ in zip(res, want[j]):
                assert a.dtype == b_.dtype and torch.equal(a, b_), (w, j)


def test_degenerate_inputs_return_or_raise(setup):
    """Inputs far from a room scan must come back well-formed or raise -- never hang, never read out of bounds."""
    cfg, model, _, _, _ = setup
    rng = np.random.default_rng(0)

    def run(xyz, cls):
        n = xyz.shape[0]
        q = np.floor(xyz / 0.02).astype(np.int32)
        uq, inv = np.unique(q, axis=0, return_inverse=True) if n else (np.zeros((0, 3), np.int32), np.zeros(0, np.int64))
        coords = np.concatenate([np.zeros((len(uq), 1), np.int32), uq], 1).astype(np.int32)
        sc = np.full((n, 20), -5.0, np.float32)
        sc[np.arange(n), cls] = 5.0
        t = {"sem_score": torch.from_numpy(sc).to(DEV), "offset": torch.zeros(n, 3, device=DEV)}
        with torch.no_grad():
            r = model(torch.zeros(len(uq), 6, device=DEV), torch.from_numpy(coords).to(DEV),
                      torch.from_numpy(xyz.astype(np.float32)).to(DEV), torch.from_numpy(inv.reshape(-1).astype(np.int64)).to(DEV),
                      None, 1, "test", teacher=t)
        torch.cuda.synchronize()
        idx, off = r["proposals"][0], r["proposals"][1]
        assert r["sem_pred_p"].shape[0] == n and off[-1].item() == idx.shape[0] and r["clt_scores"].shape[0] == off.shape[0] - 1
        assert idx.shape[0] == 0 or (0 <= int(idx[:, 1].min()) and int(idx[:, 1].max()) < n)
        return int(off.shape[0]) - 1

    with pytest.raises(ValueError):
        run(np.zeros((0, 3)), 0)
    with pytest.raises(ValueError):                                     # 700 m / 2 cm does not fit 16 bits
        run(np.array([[700.0, 0, 0], [0, 0, 0]]), 3)
    assert run(np.array([[0.1, 0.2, 0.3]]), 5) == 0                     # one point: below every class's population gate
    assert run(rng.random((500, 3)) * 0.019, 5) >= 1                    # 500 points in one voxel: one dense core
    assert run(np.tile([[0.5, 0.5, 0.5]], (500, 1)), 7) >= 1            # 500 identical points
    run(np.stack([np.linspace(0, 3, 3000), np.zeros(3000), np.zeros(3000)], 1), 4)      # a line
    assert run(rng.random((4000, 3)) * 0.5 - 2.0, 9) == 0               # sparse cloud at negative coordinates: no core
    assert run(np.concatenate([rng.random((3000, 3)) * 0.1, rng.random((3000, 3)) * 0.1 + 5.0]), 19) >= 2   # two far blobs


def test_device_front_equals_host_front_and_falls_back(setup, monkeypatch):
    """Round 5: the inference forward's gate / selection / grouping / local-scene plan on the device (two read-backs) against the
    host path (three read-backs + the cdist / topk plan in Python): bit-identical outputs on scenes with and without
    multi-entry local scenes; with a plan capacity of ONE cluster the device front reports overflow and the call is served by
    the host path -- again the same outputs."""
    import pbnet_amd.network.PBNet as PB
    cfg, model, _, _, _ = setup
    dev = "cuda:0"

    def run(b, t):
        with torch.no_grad():
            r = model(b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"], None, 1, "test", teacher=t)
        return [r["proposals"][0].cpu(), r["proposals"][1].cpu(), r["proposals"][2].cpu(), r["proposals"][3].cpu(), r["clt_scores"].cpu(),
                r["sem_pred_p"].cpu()]

    for seed, boxes, room in ((1, 6, (1.6, 1.3, 1.2)), (4, 12, (2.4, 2.0, 1.4)), (5, 2, (1.2, 1.0, 1.0))):
        b, t, _ = synth.make_val_batch(seed=seed, copies=3, room=room, n_boxes=boxes, pitch=0.03, classes=(17, 10, 5))
        b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
        t = {k: torch.from_numpy(v).to(dev) for k, v in t.items()}
        monkeypatch.setattr(PB, "DEVICE_FRONT", False)
        host = run(b, t)
        monkeypatch.setattr(PB, "DEVICE_FRONT", True)
        devf = run(b, t)
        monkeypatch.setattr(PB, "FRONT_CLUSTER_CAP", 1)
        fell = run(b, t)
        monkeypatch.setattr(PB, "FRONT_CLUSTER_CAP", 1024)
        assert host[1].shape[0] > 1, "the scene yields proposals"
        for h, d, f in zip(host, devf, fell):
            assert h.dtype == d.dtype and torch.equal(h, d)
            assert torch.equal(h, f)


def test_more_than_25_clusters_in_a_segment_take_the_reference_plan(setup, monkeypatch):
    """torch.cdist (PBNet.py:201) computes distances through a matrix multiply once a segment has more than 25
    clusters, and nearly equidistant clusters can then rank differently from the device plan's direct d^2.  Such a forward
    raises PBN_OVF_CDIST and is served by the host plan -- the reference's own cdist / topk call: a scene with ~35 instances of ONE
    class returns exactly what the host front returns, and the device plan was in fact left (the flag is read back)."""
    import pbnet_amd.network.PBNet as PB
    cfg, model, _, _, _ = setup
    dev = "cuda:0"
    b, t, _ = synth.make_val_batch(seed=9, copies=1, room=(3.2, 2.6, 1.4), n_boxes=36, pitch=0.03, classes=(5,))
    b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    t = {k: torch.from_numpy(v).to(dev) for k, v in t.items()}

    def run():
        with torch.no_grad():
            r = model(b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"], None, 1, "test", teacher=t, n_batch=1)
        return [r["proposals"][0].cpu(), r["proposals"][1].cpu(), r["clt_scores"].cpu()]
    seen = []
    orig = PB.PBNet._device_front

    def spy(self, *a, **k):
        out = orig(self, *a, **k)
        seen.append(out is None)
        return out
    monkeypatch.setattr(PB.PBNet, "_device_front", spy)
    monkeypatch.setattr(PB, "DEVICE_FRONT", True)
    got = run()
    monkeypatch.setattr(PB, "DEVICE_FRONT", False)
    want = run()
    assert seen and seen[0], "the device front did not report the > 25-cluster segment"
    assert want[1].shape[0] - 1 >= 26, "the scene yields more than 25 proposals of one class (%d)" % (want[1].shape[0] - 1)
    for g, w in zip(got, want):
        assert torch.equal(g, w)
