"""Training step (BASELINE configs[2], one rank's share): model_fn forward + losses + backward on one ScanNet-sized
synthetic scene, bf16 slabs / fp32 master weights, teacher-forced heads so that the cluster stage is active."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet, model_fn
dev = torch.device("cuda", 0)
dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[os.environ.get("PBN_TRAIN_DTYPE", "bf16")]
cfg = get_config(batch_size=1, cluster_epoch=0)
torch.manual_seed(22)
model = PBNet(cfg).to(dev).train()
batch_np, teacher_np, info_ = synth.make_val_batch(seed=10, copies=1)
t = torch.from_numpy
batch = {k: t(v) for k, v in batch_np.items()}
batch["feat_voxel"] = batch["feat_voxel"].to(dtype)
n = batch["xyz_original"].shape[0]
ins = batch["ins"]
n_inst = int(ins.max().item()) + 1
sem = teacher_np["sem_score"].argmax(1)
info = torch.zeros(n, 9)
pointnum = []
for i in range(n_inst):
    m = ins == i
    pointnum.append(int(m.sum()))
    if m.any():
        info[m, 0:3] = batch["xyz_original"][m].mean(0)
batch.update(sem=t(sem).long(), inst_info=info, instance_pointnum=torch.tensor(pointnum, dtype=torch.int32))
batch = {k: v.to(dev) for k, v in batch.items()}
teacher = {k: t(v).to(dev) for k, v in teacher_np.items()}
orig_forward = model.forward
model.forward = lambda *a, **kw: orig_forward(*a, teacher=teacher, **kw)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
def step():
    opt.zero_grad(set_to_none=True)
    loss, pred, visual, meter = model_fn(batch, model, 1, cfg, "train")
    loss.backward()
    opt.step()
    return loss
for _ in range(2): step()
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K): loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("train step (%d pts, %d voxels, %s slabs): %.1f ms  -> %.2f scenes/s per GPU; loss %.4f"
      % (info_["n_points"], info_["n_voxels"], dtype, dt * 1e3, 1 / dt, float(loss)))
