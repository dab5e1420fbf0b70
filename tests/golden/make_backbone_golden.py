"""Generates tests/golden/backbone_*.npz (SURVEY.md 8c fixtures B1-B5) with the CPU oracle (oracle/sparse_ref.py).

MinkowskiEngine is absent, so the expected outputs are the oracle's: PARITY UNPINNED w.r.t. upstream ME.  Weights are
not stored (tens of MB): they are re-created from torch.manual_seed(22) (the reference's seed, config/config.py:15) on
the CPU generator, which is bit-reproducible for a fixed torch version; the fixture records torch.__version__.

Run from the repo root:  python tests/golden/make_backbone_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import sparse_ref as R  # noqa: E402
from pbnet_amd import synth  # noqa: E402
from pbnet_amd.network.Mink import Mink_unet  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def small_scene(seed, n_target=2000):
    sc = synth.synth_room(seed=seed, pitch=0.0225, room=(0.6, 0.5, 0.4), n_boxes=1)
    q, first, inv = synth.voxelize_numpy(sc["xyz"], 0.02)
    coords = np.concatenate([np.zeros((len(q), 1), np.int32), q], 1).astype(np.int32)
    return coords


def build(arch, cin, seed=22):
    torch.manual_seed(seed)
    m = Mink_unet(cin, 32, arch=arch)
    # non-trivial BN statistics / affine so the eval path is actually exercised
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
    return m


def main():
    for arch, cin, sseed in (("MinkUNet14A", 34, 31), ("MinkUNet34C", 6, 32)):
        coords = small_scene(sseed)
        # second batch element: the same scene shifted (exercises batch separation in the hash)
        c2 = coords.copy()
        c2[:, 0] = 1
        c2[:, 1:] += np.array([3, -2, 1], np.int32)
        coords = np.concatenate([coords, c2[: len(c2) // 2]], 0)
        g = torch.Generator().manual_seed(7)
        feats = torch.randn(len(coords), cin, generator=g)
        m = build(arch, cin)
        sd = m.state_dict()
        out_eval = R.minkunet_forward(sd, arch, feats, coords, training=False)
        out_train = R.minkunet_forward(sd, arch, feats, coords, training=True)
        cm = R.CoordinateManager(coords)
        counts = np.array([cm.get_coords(s).shape[0] for s in (1, 2, 4, 8, 16)], np.int32)
        pairs = np.array([cm.n_pairs(s, s, 3) for s in (1, 2, 4, 8, 16)], np.int64)
        np.savez_compressed(os.path.join(OUT, "backbone_%s.npz" % arch), coords=coords, feats=feats.numpy(),
                            out_eval=out_eval.numpy(), out_train=out_train.numpy(), counts=counts, pairs=pairs,
                            torch_version=np.array(torch.__version__), seed=np.int32(22))
        print(arch, len(coords), counts.tolist(), pairs.tolist(), float(out_eval.abs().mean()), float(out_train.abs().mean()))


if __name__ == "__main__":
    main()
