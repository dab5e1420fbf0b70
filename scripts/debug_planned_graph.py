import torch, sys, os
sys.path.insert(0, ".")
from pbnet_amd import planned, synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet
DEV="cuda:0"
cfg = get_config(test=True); torch.manual_seed(22); model = PBNet(cfg)
if os.environ.get("BNRAND", "1") != "0":
    g = torch.Generator().manual_seed(5)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
model = model.to(DEV).eval()
batch, teacher, info = synth.make_val_batch(seed=1, copies=3, room=(1.6, 1.3, 1.2), n_boxes=6, pitch=0.03, classes=(17, 10))
b = {k: torch.from_numpy(v).to(DEV) for k, v in batch.items()}; b["feat_voxel"] = b["feat_voxel"].half()
t = {k: torch.from_numpy(v).to(DEV) for k, v in teacher.items()}
args = (b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"])
cap = planned.measure_capacities(model, *args, teacher=t).padded(1.3)
if os.environ.get("ZERO"):
    _empty = torch.empty
    def zempty(*a, **k):
        x = _empty(*a, **k)
        if x.numel():
            x.view(torch.uint8).fill_(int(os.environ["ZERO"]) & 255) if x.is_contiguous() else None
        return x
    planned.torch.empty = zempty
pf = planned.PlannedForward(model, cap, dtype=torch.float16)
want = pf(*args, teacher=t)
pf.capture(*args, teacher=t)
got = pf.finish(pf.replay())
print("ZERO", os.environ.get("ZERO"), "FOLD", os.environ.get("PBNET_FOLD_SHORTCUT"), "FAMILY", os.environ.get("PBN_CONV_FAMILY"),
      [bool(torch.equal(a, w)) for a, w in zip(got["proposals"], want["proposals"])], bool(torch.equal(got["clt_scores"], want["clt_scores"])))
got2 = pf.finish(pf.replay())
want2 = pf(*args, teacher=t)
print("replay vs replay", bool(torch.equal(got2["proposals"][3], got["proposals"][3])), "eager vs eager", bool(torch.equal(want2["proposals"][3], want["proposals"][3])))
d = (got["proposals"][3].float() - want["proposals"][3].float()).abs()
print("   diff max %g count %d of %d" % (d.max().item(), int((d > 0).sum()), d.numel()))
