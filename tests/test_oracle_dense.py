"""CPU tier: oracle/sparse_ref.py against the independent DENSE statement of the backbone (tests/dense_backbone.py:
F.conv3d / F.conv_transpose3d / batch_norm on a densified grid, no coordinate or kernel maps).  Narrows the
"parity unpinned" gap of the MinkowskiEngine boundary: the oracle's conventions C1-C5, C7, C8 (offset order, weight
layout, even kernels not centred, transposed map, BN on active rows) agree with the dense operators MinkowskiEngine
generalises.  Layers of /root/reference/network/Mink.py:221-288, wiring :291-354."""
import numpy as np
import torch

import dense_backbone as D
from oracle import sparse_ref as R
from pbnet_amd import synth

TOL = 1e-4


def _coords(seed, extent, batch=2, fill=0.12, lo=-16):
    """Random occupancy + a planar sheet, negative coordinates included; unique rows."""
    rng = np.random.default_rng(seed)
    parts = []
    for b in range(batch):
        n = int(fill * extent ** 3)
        c = rng.integers(lo, lo + extent, (n, 3))
        xy = np.stack(np.meshgrid(np.arange(lo, lo + extent), np.arange(lo, lo + extent), indexing="ij"), -1).reshape(-1, 2)
        sheet = np.concatenate([xy, np.full((len(xy), 1), lo + extent // 2 + b)], 1)
        c = np.unique(np.concatenate([c, sheet], 0), axis=0)
        c = c[rng.permutation(len(c))]
        parts.append(np.concatenate([np.full((len(c), 1), b), c], 1))
    return np.concatenate(parts, 0).astype(np.int32)


def _bn_sd(prefix, C, g):
    return {prefix + ".bn.weight": torch.rand(C, generator=g) + 0.5, prefix + ".bn.bias": torch.randn(C, generator=g) * 0.1,
            prefix + ".bn.running_mean": torch.randn(C, generator=g) * 0.1,
            prefix + ".bn.running_var": torch.rand(C, generator=g) * 0.5 + 0.75}


def _close(got, want, what):
    err = (got - want).abs().max().item()
    print("%s: max |diff| %.3e (scale %.2f)" % (what, err, want.abs().max().item()))
    assert err <= TOL, (what, err)


def test_single_convolutions_k3_k5_1x1():
    g = torch.Generator().manual_seed(1)
    coords = _coords(3, 32)
    sc = D.DenseScene(coords, 32)
    cm = R.CoordinateManager(coords)
    n = len(coords)
    for K, ci, co in ((3, 8, 16), (5, 6, 32), (3, 32, 8)):
        x = torch.randn(n, ci, generator=g)
        w = torch.randn(K ** 3, ci, co, generator=g) / np.sqrt(ci * 9)
        want = sc.sample(D.conv_same(sc.scatter(x), w, K))
        got = R.conv(x, w, cm.get_map(1, 1, K), n)
        _close(got, want, "k=%d %d->%d" % (K, ci, co))
    x = torch.randn(n, 24, generator=g)
    w = torch.randn(24, 16, generator=g) / 5
    _close(R.conv(x, w, None, n), sc.sample(D.conv_same(sc.scatter(x), w, 1)), "1x1")


def test_strided_down_and_transposed_up_all_levels():
    g = torch.Generator().manual_seed(2)
    coords = _coords(4, 32, fill=0.05)
    sc = D.DenseScene(coords, 32)
    shift = sc.c[0, 1:] - coords[0, 1:].astype(np.int64)
    cm = R.CoordinateManager(coords)
    x = torch.randn(len(coords), 8, generator=g)
    grid = sc.scatter(x)
    for s in (1, 2, 4, 8):
        w = torch.randn(8, 8, 8, generator=g) / 8
        fine, coarse = cm.get_coords(s), cm.get_coords(2 * s)
        assert int(sc.mask[2 * s].sum()) == len(coarse), "strided coordinate set == max-pooled occupancy"
        # down: oracle rows vs dense conv3d(stride 2) sampled at the oracle's coarse coordinates
        got = R.conv(x, w, cm.get_map(s, 2 * s, 2), len(coarse))
        dn = torch.nn.functional.conv3d(grid, D._w3(w, 2), stride=2)
        cc = coarse.astype(np.int64).copy()
        cc[:, 1:] += shift[None, :]
        _close(got, sc.rows_at(dn, cc, 2 * s), "k2s2 down @%d" % s)
        # up: the coarse features back onto the fine set
        wt = torch.randn(8, 8, 8, generator=g) / 8
        got_up = R.conv_transpose(got, wt, cm.get_map(s, 2 * s, 2), len(fine))
        upg = torch.nn.functional.conv_transpose3d(dn * sc.mask[2 * s], D._w3t(wt, 2), stride=2)
        fc = fine.astype(np.int64).copy()
        fc[:, 1:] += shift[None, :]
        _close(got_up, sc.rows_at(upg, fc, s), "k2s2 transposed up @%d" % (2 * s))
        x, grid = got, dn * sc.mask[2 * s]


def test_basic_block_with_shortcut_eval_and_train():
    g = torch.Generator().manual_seed(3)
    coords = _coords(5, 32)
    sc = D.DenseScene(coords, 32)
    cm = R.CoordinateManager(coords)
    sd = {"b.conv1.kernel": torch.randn(27, 24, 32, generator=g) / 15, "b.conv2.kernel": torch.randn(27, 32, 32, generator=g) / 17,
          "b.downsample.0.kernel": torch.randn(24, 32, generator=g) / 5}
    for p, C in (("b.norm1", 32), ("b.norm2", 32), ("b.downsample.1", 32)):
        sd.update(_bn_sd(p, C, g))
    x = torch.randn(len(coords), 24, generator=g)
    for training in (False, True):
        want = sc.sample(D.basic_block_dense(sc.scatter(x), sc.mask[1], sd, "b", training))
        got = R.basic_block(x, sd, "b", cm, 1, training)
        _close(got, want, "BasicBlock+shortcut training=%s" % training)


def _net_case(arch, cin, extent, seed):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_backbone_golden as G
    coords = _coords(seed, extent, fill=0.04)
    net = G.build(arch, cin)
    sd = net.state_dict()
    feats = torch.randn(len(coords), cin, generator=torch.Generator().manual_seed(seed))
    return sd, coords, feats


def test_whole_minkunet14a_dense_vs_oracle():
    sd, coords, feats = _net_case("MinkUNet14A", 34, 48, 6)
    for training in (False, True):
        want = D.minkunet_dense(sd, "MinkUNet14A", feats, coords, 48, training)
        got = R.minkunet_forward(sd, "MinkUNet14A", feats, coords, training=training)
        _close(got, want, "MinkUNet14A %d voxels training=%s" % (len(coords), training))


def test_whole_minkunet34c_dense_vs_oracle():
    sd, coords, feats = _net_case("MinkUNet34C", 6, 32, 7)
    want = D.minkunet_dense(sd, "MinkUNet34C", feats, coords, 32, False)
    got = R.minkunet_forward(sd, "MinkUNet34C", feats, coords, training=False)
    _close(got, want, "MinkUNet34C %d voxels eval" % len(coords))


def test_synthetic_room_surface_scene():
    """The bench generator's geometry (thin surfaces), one BasicBlock stack at stride 2 and 4 through the real pyramid."""
    sc0 = synth.synth_room(seed=41, pitch=0.0225, room=(0.6, 0.5, 0.4), n_boxes=1)
    q, _, _ = synth.voxelize_numpy(sc0["xyz"], 0.02)
    coords = np.concatenate([np.zeros((len(q), 1), np.int32), q - 7], 1).astype(np.int32)   # negative coordinates too
    sd, _, _ = _net_case("MinkUNet14A", 34, 48, 8)
    feats = torch.randn(len(coords), 34, generator=torch.Generator().manual_seed(9))
    taps = {}
    want = D.minkunet_dense(sd, "MinkUNet14A", feats, coords, 48, False, taps=taps)
    rt = {}
    got = R.minkunet_forward(sd, "MinkUNet14A", feats, coords, taps=rt)
    _close(rt["out_p1"], taps["out_p1"], "stem k=5 on the room scene")
    _close(got, want, "MinkUNet14A on the room scene (%d voxels)" % len(coords))
