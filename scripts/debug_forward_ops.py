"""Which torch ops (and from where) the inference forward still issues: torch profiler with Python stacks, bench scene."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet
dev = torch.device("cuda", 0)
cfg = get_config(test=True)
torch.manual_seed(22)
model = PBNet(cfg).to(dev).eval()
batch, teacher, info = synth.make_val_batch(seed=2, copies=1)
b = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
b["feat_voxel"] = b["feat_voxel"].to(torch.bfloat16)
t = {k: torch.from_numpy(v).to(dev) for k, v in teacher.items()}


def step():
    with torch.no_grad():
        return model(b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"], None, 1, "test", teacher=t)


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.autograd.profiler.profile(with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
c = collections.Counter()
for ev in prof.function_events:
    if ev.name.startswith("aten::") and ev.cpu_parent is None:
        where = [s for s in (ev.stack or []) if "pbnet_amd" in s][:1]
        c[(ev.name, str(ev.input_shapes)[:60], where[0].split("pbnet_amd/")[-1][:60] if where else "")] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
print("top-level aten ops per forward:", sum(c.values()))
