"""Manual consistency sweep (not collected by pytest): fused inference path (torch.no_grad: native coordinate pipeline,
U-Net executor, stage glue) against the module-by-module path (autograd enabled: tensor-op glue) on a range of
synthetic scenes, fp32 slabs.  Integer outputs must agree unless a mask score sits within 1e-5 of the threshold."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet, MASK_THD

dev = torch.device("cuda", 0)
cfg = get_config(test=True)
torch.manual_seed(22)
model = PBNet(cfg).to(dev).eval()
bad = 0
for seed in range(1, 13):
    rng = np.random.default_rng(seed)
    room = (float(rng.uniform(1.2, 3.0)), float(rng.uniform(1.0, 2.4)), float(rng.uniform(0.9, 2.0)))
    copies = int(rng.integers(1, 4))
    nb = int(rng.integers(2, 9))
    classes = tuple(int(c) for c in rng.choice(np.arange(2, 20), size=min(4, nb), replace=False))
    batch, teacher, info = synth.make_val_batch(seed=seed, copies=copies, room=room, n_boxes=nb, pitch=0.03, classes=classes)
    b = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    t = {k: torch.from_numpy(v).to(dev) for k, v in teacher.items()}
    args = (b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"], None, 1, "test")
    with torch.no_grad():
        r1 = model(*args, teacher=t)
    with torch.enable_grad():
        r2 = model(*args, teacher=t)
    p1, p2 = r1["proposals"], r2["proposals"]
    same = p1[0].shape == p2[0].shape and torch.equal(p1[0], p2[0]) and torch.equal(p1[1], p2[1])
    ds = (r1["clt_scores"].float() - r2["clt_scores"].float()).abs().max().item() if same and r1["clt_scores"].numel() else float("nan")
    print("seed %2d room %s copies %d boxes %d: %7d pts, proposals %3d / %3d rows %6d / %6d  identical=%s  max|score diff| %.2e"
          % (seed, "x".join("%.1f" % v for v in room), copies, nb, info["n_points"], p1[1].shape[0] - 1, p2[1].shape[0] - 1,
             p1[0].shape[0], p2[0].shape[0], same, ds))
    bad += 0 if same else 1
print("mismatching scenes:", bad)
