"""CPU-side time per autograd op of the training step's backward (torch profiler, small scene)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet, model_fn
dev = torch.device("cuda", 0)
cfg = get_config(batch_size=1, cluster_epoch=0)
torch.manual_seed(22)
model = PBNet(cfg).to(dev).train()
small = "--full" not in sys.argv
kw = dict(room=(1.6, 1.3, 1.2), n_boxes=4, pitch=0.03, classes=(17, 10)) if small else {}
batch_np, teacher_np, info = synth.make_train_batch(seed=10, copies=1, **kw)
t = torch.from_numpy
batch = {k: t(v).to(dev) for k, v in batch_np.items()}
batch["feat_voxel"] = batch["feat_voxel"].to(torch.bfloat16)
teacher = {k: t(v).to(dev) for k, v in teacher_np.items()}
fwd = model.forward
model.forward = lambda *a, **k: fwd(*a, teacher=teacher, **k)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
for _ in range(3):
    opt.zero_grad(set_to_none=True)
    loss = model_fn(batch, model, 1, cfg, "train")[0]
    loss.backward()
    opt.step()
torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
loss = model_fn(batch, model, 1, cfg, "train")[0]
torch.cuda.synchronize()
with torch.autograd.profiler.profile(with_stack="--stack" in sys.argv, record_shapes=True) as prof:
    loss.backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=60))
if "--stack" in sys.argv:
    for ev in prof.function_events:
        if ev.name == "aten::copy_" and ev.stack:
            print("copy_ from:", [s for s in ev.stack if "pbnet_amd" in s or "scripts" in s][:3])
            break
    import collections
    c = collections.Counter()
    for ev in prof.function_events:
        if ev.name == "aten::copy_":
            c[tuple(s for s in (ev.stack or []) if "pbnet_amd" in s)[:2]] += 1
    for k, v in c.most_common(6):
        print(v, k)
    c = collections.Counter()
    for ev in prof.function_events:
        if ev.name == "aten::copy_":
            c[str(ev.input_shapes)] += 1
    for k, v in c.most_common(12):
        print(v, k)
