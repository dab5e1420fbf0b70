"""CPU tier: the loss arithmetic of model_fn (/root/reference/network/PBNet.py:366-416, diceLoss :463-468) --
pbnet_amd.network.PBNet.model_losses against the independent numpy restatement oracle/loss_ref.py, 1e-5 per term,
incl. the reference's in-place `gt_mask[gt_mask == -1.] = 0.5` on a long tensor (ignore rows become target 0, the
dice term covers every row, the mutated mask is returned)."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import model_losses, get_segmented_scores

TOL = 1e-5


def _case(seed, n=5000, n_inst=7, rows=3000, n_prop=9):
    rng = np.random.default_rng(seed)
    ins = rng.integers(0, n_inst, n)
    ins[rng.random(n) < 0.3] = -100
    sem = rng.integers(0, 20, n)
    sem[rng.random(n) < 0.1] = -100
    xyz = rng.uniform(0, 4, (n, 3)).astype(np.float32)
    info = np.zeros((n, 9), np.float32)
    for i in range(n_inst):
        m = ins == i
        if m.any():
            info[m, :3] = xyz[m].mean(0)
    pointnum = np.array([(ins == i).sum() for i in range(n_inst)], np.int32)
    sem_score = rng.normal(0, 2, (n, 20)).astype(np.float32)
    offset = rng.normal(0, 0.3, (n, 3)).astype(np.float32)
    offset[:5] = 0.0                                             # zero-norm predictions: the +1e-8 guards
    pred_mask = rng.uniform(0, 1, rows).astype(np.float32)
    pred_mask[:3] = [0.0, 1.0, 1e-30]                            # log clamp at -100
    gt_mask = rng.integers(0, 2, rows).astype(np.int64)
    gt_mask[rng.random(rows) < 0.2] = -1
    gt_mask[:3] = [1, 0, 1]
    lens = rng.integers(20, 400, n_prop)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pidx = np.concatenate([rng.choice(n, l, replace=False) for l in lens]).astype(np.int64)
    # make some proposals align with instances so IoUs span the fg / bg thresholds
    for p in range(0, n_prop, 2):
        members = np.nonzero(ins == (p % n_inst))[0]
        take = members[: lens[p]]
        pidx[off[p]:off[p] + len(take)] = take
    clt = rng.uniform(0.01, 0.99, n_prop).astype(np.float32)
    return dict(ins=ins, sem=sem, xyz=xyz, info=info, pointnum=pointnum, sem_score=sem_score, offset=offset,
                pred_mask=pred_mask, gt_mask=gt_mask, pidx=pidx, off=off, clt=clt)


def _torch_iou(pidx, off, ins, pointnum):
    return torch.from_numpy(loss_ref.get_iou(pidx.numpy(), off.numpy(), ins.numpy(), pointnum.numpy()))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_losses_match_restatement(seed):
    c = _case(seed)
    t = torch.from_numpy
    cfg = get_config(cluster_epoch=0)
    gt_mask_t = t(c["gt_mask"].copy())
    ret = {"sem_pred_score_p": t(c["sem_score"]), "offset_pred_p": t(c["offset"]),
           "mask_scores": (t(c["pred_mask"]).view(-1, 1), gt_mask_t),
           "proposals": (torch.stack([torch.zeros(len(c["pidx"]), dtype=torch.int64), t(c["pidx"])], 1), t(c["off"]), None, None),
           "clt_scores": t(c["clt"])}
    loss, parts, valid, weight, gm = model_losses(ret, t(c["sem"]), t(c["ins"]), t(c["info"]), t(c["pointnum"]), t(c["xyz"]),
                                                  1, cfg, get_iou=_torch_iou)
    want = loss_ref.losses(c["sem_score"], c["offset"], c["sem"], c["ins"], c["info"], c["xyz"],
                           mask=(c["pred_mask"], c["gt_mask"]), proposals=(c["pidx"], c["off"]), clt_scores=c["clt"],
                           instance_pointnum=c["pointnum"], fg=cfg.fg_thresh, bg=cfg.bg_thresh)
    for k in ("semantic_loss", "offset_norm_loss", "offset_dir_loss", "mask_loss", "dice_loss", "score_loss", "loss"):
        got = float(parts[k])
        print("%s: %.7f vs %.7f" % (k, got, want[k]))
        assert abs(got - want[k]) <= TOL * max(1.0, abs(want[k])), k
    # the mutated mask is what travels on (PBNet.py:398, :437): no -1 left, ignore rows are 0
    assert gm is gt_mask_t and int((gm == -1).sum()) == 0
    assert np.array_equal(gm.numpy(), want["gt_mask"])
    assert float(weight.sum()) == float((c["gt_mask"] != -1).sum())
    # below cluster_epoch only the three point-wise terms exist
    cfg2 = get_config(cluster_epoch=128)
    loss2, parts2, _, w2, _ = model_losses(ret, t(c["sem"]), t(c["ins"]), t(c["info"]), t(c["pointnum"]), t(c["xyz"]), 1, cfg2)
    assert set(parts2) == {"semantic_loss", "offset_norm_loss", "offset_dir_loss", "loss"} and w2 is None
    assert abs(float(loss2) - (want["semantic_loss"] + want["offset_norm_loss"] + want["offset_dir_loss"])) <= 1e-5


def test_dice_covers_ignore_rows_as_zero_targets():
    """The advisor's case: a mask with -1 entries.  Dropping the ignore rows from the dice term (round 1) gives a
    different number than the reference's all-rows form."""
    p = np.array([0.9, 0.8, 0.7, 0.2], np.float32)
    g = np.array([1, -1, -1, 0], np.int64)
    want = loss_ref.losses(np.zeros((1, 20), np.float32), np.zeros((1, 3), np.float32), np.array([0]), np.array([0]),
                           np.zeros((1, 9), np.float32), np.zeros((1, 3), np.float32), mask=(p, g),
                           proposals=(np.array([0]), np.array([0, 1])), clt_scores=np.array([0.5], np.float32),
                           instance_pointnum=np.array([1], np.int32))
    all_rows = 1 - (2 * float(p[0]) + 1) / (1 + (p.astype(np.float64) ** 2).sum() + 1 + 1e-8)
    kept_rows = 1 - (2 * 0.9 + 1) / (1 + 0.81 + 0.04 + 1 + 1e-8)
    assert abs(want["dice_loss"] - all_rows) < 1e-6 and abs(all_rows - kept_rows) > 0.05
    from pbnet_amd.network.PBNet import diceLoss
    gm = torch.from_numpy(g.copy())
    gm[gm == -1] = 0
    assert abs(float(diceLoss(torch.from_numpy(p), gm)) - all_rows) < 1e-6


def test_segmented_scores_pinned_by_reference_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "segmented_scores.npz"))
    for key, (fg, bg) in (("fg0.75_bg0.25", (0.75, 0.25)), ("fg1_bg0", (1.0, 0.0)), ("fg0.5_bg0.2", (0.5, 0.2))):
        assert np.array_equal(loss_ref.segmented_scores(d["scores"], fg, bg), d[key]), key
        assert np.array_equal(get_segmented_scores(torch.from_numpy(d["scores"]), fg, bg).numpy(), d[key]), key
