import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pbnet_amd import planned, synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet
DEV = "cuda:0"
cfg = get_config(test=True)
torch.manual_seed(22)
model = PBNet(cfg).to(DEV).eval()
which = sys.argv[1]
kw = dict(seed=2, copies=1) if which == "bench" else dict(seed=1, copies=int(which), room=(1.6, 1.3, 1.2), n_boxes=6, pitch=0.03, classes=(17, 10))
batch, teacher, info = synth.make_val_batch(**kw)
b = {k: torch.from_numpy(v).to(DEV) for k, v in batch.items()}
b["feat_voxel"] = b["feat_voxel"].to(torch.bfloat16)
t = {k: torch.from_numpy(v).to(DEV) for k, v in teacher.items()}
args = (b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"])
cap = planned.measure_capacities(model, *args, teacher=t).padded(1.25)
pf = planned.PlannedForward(model, cap, dtype=torch.bfloat16)
want = pf(*args, teacher=t)
pf.capture(*args, teacher=t)
print(which, "captured", flush=True)
for i in range(3):
    t0 = time.perf_counter()
    out = pf.replay()
    torch.cuda.synchronize()
    print(which, "replay", i, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
got = pf.finish(out)
print(which, "equal:", all(torch.equal(a, w) for a, w in zip(got["proposals"], want["proposals"])), flush=True)
