"""GPU tier: BASELINE.json configs[0] at its stated workload (SURVEY.md 8d "C1"): ONE un-rotated synthetic scene,
``synth.synth_room(seed=1, pitch=0.0225, room=(1.6, 1.3, 1.2), n_boxes=1)`` at 2 cm voxels, MinkUNet14A (6 -> 32) in fp32
plus the PB_lib grouping call -- the reference's own CPU-runnable plumbing case (network/Mink.py:357-383,502-526;
lib/PB_lib/torch_io/pbnet_ops.py:14-82), here on the MI355X path against the CPU oracle:

  * ``ME.utils.sparse_quantize`` / ``sparse_collate`` build the network input exactly as dataset_preprocess.py:269-296 does;
  * backbone features of every voxel and of every point (``out.F[inverse]``): |diff| <= 1e-4 ABSOLUTE vs oracle/sparse_ref.py;
  * ``pbnet_ops.cluster`` (the reference-shaped op, CPU tensors in and out) on the teacher-forced box class and on the whole
    foreground as one mixed call: every integer output and the centre BITS equal oracle/pb_cluster_ref.c.
"""
import numpy as np
import pytest
import torch

from oracle import pb_cluster_ref as cluster_oracle
from oracle import sparse_ref as R
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import pbnet_ops, synth
from pbnet_amd.network.Mink import Mink_unet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4
VOXEL = 0.02
RADIUS, MIN_PTS = 0.04, 31          # config/config.py:44-45


@pytest.fixture(scope="module")
def c1():
    sc = synth.synth_room(seed=1, pitch=0.0225, room=(1.6, 1.3, 1.2), n_boxes=1)
    q, first, inv = synth.voxelize_numpy(sc["xyz"], VOXEL)
    feats = np.concatenate([sc["rgb"], sc["normal"]], 1).astype(np.float32)
    print("C1: %d points, %d voxels @ %.0f mm" % (len(sc["xyz"]), len(q), VOXEL * 1e3))
    assert 20000 <= len(sc["xyz"]) <= 26000 and 18000 <= len(q) <= 22000       # "1 synthetic scene, 20k pts"
    return sc, q, first, inv, feats


def test_c1_loader_ops_build_the_same_input(c1):
    sc, q, first, inv, feats = c1
    qc, qf, index, inverse = ME.utils.sparse_quantize(sc["xyz"], feats, quantization_size=VOXEL, return_index=True,
                                                      return_inverse=True)
    assert np.array_equal(qc, q) and np.array_equal(index, first) and np.array_equal(inverse, inv)
    assert np.array_equal(qf, feats[first])
    bc, bf = ME.utils.sparse_collate([torch.from_numpy(qc)], [torch.from_numpy(qf)])
    want = R.batched_coordinates([q.astype(np.float32)])
    assert bc.dtype == torch.int32 and np.array_equal(bc.numpy(), want)
    assert torch.equal(bf, torch.from_numpy(feats[first]))


def test_c1_minkunet14a_fp32(c1):
    sc, q, first, inv, feats = c1
    coords = np.concatenate([np.zeros((len(q), 1), np.int32), q], 1).astype(np.int32)
    x = torch.from_numpy(feats[first])
    torch.manual_seed(22)                                  # config/config.py:15
    net = Mink_unet(6, 32, arch="MinkUNet14A")
    g = torch.Generator().manual_seed(5)
    for mod in net.modules():                              # non-trivial BN statistics: the folded epilogue is exercised
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    want = R.minkunet_forward(sd, "MinkUNet14A", x, coords)
    net = net.to(DEV).eval()
    with torch.no_grad():
        out = net(ME.SparseTensor(x, torch.from_numpy(coords), device=DEV))
        got_v = out.F.float().cpu()
        got_p = out.F[torch.from_numpy(inv).to(DEV)].float().cpu()        # voxel -> point (PBNet.py:130)
    err_v = (got_v - want).abs().max().item()
    err_p = (got_p - want[torch.from_numpy(inv)]).abs().max().item()
    print("MinkUNet14A fp32 on %d voxels: max |diff| %.2e (voxels), %.2e (points), scale %.2f" %
          (len(q), err_v, err_p, want.abs().max().item()))
    assert got_v.shape == (len(q), 32)
    assert err_v <= TOL and err_p <= TOL


def _cluster_both(off, org, sem, seg):
    t = torch.from_numpy
    cid, cnum, den, center = pbnet_ops.cluster(t(off), t(org), t(sem.astype(np.int32)), t(seg.astype(np.int32)), RADIUS,
                                               MIN_PTS, len(seg))
    assert cid.device.type == "cpu" and cid.dtype == torch.int32 and center.dtype == torch.float32
    w_id, w_num, w_den, w_center = cluster_oracle.cluster(off, org, sem, seg, RADIUS, MIN_PTS)
    assert np.array_equal(cid.numpy(), w_id), "cluster_id"
    assert np.array_equal(cnum.numpy(), w_num), "cluster_num"
    assert np.array_equal(den.numpy(), w_den), "den_queue (+1)"
    assert np.array_equal(center.numpy().view(np.int32), w_center.view(np.int32)), "centre bits"
    return w_id, w_num


def test_c1_grouping_box_class(c1):
    sc = c1[0]
    sem_pred, offset = synth.teacher_forced_heads(sc, seed=1)
    sel = np.nonzero(sem_pred == 2)[0]                     # the box (class 2) + the 2 % flips that landed on it
    org = sc["xyz"][sel]
    off = (org + offset[sel]).astype(np.float32)
    ids, num = _cluster_both(off, org, sem_pred[sel], np.array([len(sel)]))
    print("class 2: %d points -> %d clusters, %d unassigned" % (len(sel), int(num.sum()), int((ids < 0).sum())))
    assert int(num.sum()) >= 1


def test_c1_grouping_mixed_call_three_segments(c1):
    """The general rule (component x class) on the same scene: every foreground point in one call,
    split into three batch segments of which the middle one is empty (cluster.cu:59-61)."""
    sc = c1[0]
    sem_pred, offset = synth.teacher_forced_heads(sc, seed=1)
    sel = np.nonzero(sem_pred >= 2)[0]
    org = sc["xyz"][sel]
    off = (org + offset[sel]).astype(np.float32)
    half = len(sel) // 2
    ids, num = _cluster_both(off, org, sem_pred[sel], np.array([half, 0, len(sel) - half]))
    assert num[1] == 0 and int(num.sum()) >= 2
