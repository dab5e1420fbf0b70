"""cProfile of the host side of one configs[2] training step (forward and backward separately), after warm-up:
where the Python time of the step goes once the U-Net bodies run on the native executor."""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet, model_fn

dev = torch.device("cuda", 0)
cfg = get_config(batch_size=1, cluster_epoch=0)
torch.manual_seed(22)
model = PBNet(cfg).to(dev).train()
batch_np, teacher_np, info = synth.make_train_batch(seed=10, copies=1)
t = torch.from_numpy
batch = {k: t(v).to(dev) for k, v in batch_np.items()}
batch["feat_voxel"] = batch["feat_voxel"].to(torch.bfloat16)
teacher = {k: t(v).to(dev) for k, v in teacher_np.items()}
fwd = model.forward
model.forward = lambda *a, **k: fwd(*a, teacher=teacher, **k)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
for _ in range(3):
    opt.zero_grad(set_to_none=True)
    loss, *_ = model_fn(batch, model, 1, cfg, "train")
    loss.backward(); opt.step()
torch.cuda.synchronize()
for phase in ("forward", "backward"):
    pr = cProfile.Profile()
    opt.zero_grad(set_to_none=True)
    if phase == "forward":
        pr.enable(); loss, *_ = model_fn(batch, model, 1, cfg, "train"); pr.disable()
        loss.backward()
    else:
        loss, *_ = model_fn(batch, model, 1, cfg, "train")
        torch.cuda.synchronize()
        pr.enable(); loss.backward(); pr.disable()
    opt.step(); torch.cuda.synchronize()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        print("==== %s by %s" % (phase, key))
        print("\n".join(l[:170] for l in s.getvalue().splitlines()[4:]))
