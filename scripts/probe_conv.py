"""Per-layer timing of the fused MinkUNet34C forward on the S150 scene (HIP events around each conv launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine import conv as C
from pbnet_amd.network import mink_unet as U
from pbnet_amd.network.Mink import Mink_unet

dev = "cuda:0"
dt = {"bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
batch, teacher, info = synth.make_val_batch(seed=2, copies=1)
coords = torch.from_numpy(batch["xyz_voxel"]).to(dev)
feats = torch.from_numpy(batch["feat_voxel"]).to(dev).to(dt)
torch.manual_seed(22)
net = Mink_unet(6, 32, arch="MinkUNet34C").to(dev).eval()
x = ME.SparseTensor(feats, coords)
with torch.no_grad():
    for _ in range(3): net(x)
recs = []
orig = C.spconv_forward
def wrapped(feats, nbr, n_out, packed, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(feats, nbr, n_out, packed, **kw); e1.record()
    k = 1 if nbr is None else nbr.shape[1]
    recs.append((e0, e1, int(n_out), int(feats.shape[1]), packed[3], k))
    return out
C.spconv_forward = wrapped; U.spconv_forward = wrapped
with torch.no_grad():
    for _ in range(5): net(x)
torch.cuda.synchronize()
n = len(recs) // 5
tot = 0
agg = {}
for i in range(n):
    ts = [recs[j * n + i][0].elapsed_time(recs[j * n + i][1]) for j in range(5)]
    t = sorted(ts)[2] * 1e3
    _, _, rows, cin, cout, k = recs[i]
    tot += t
    key = (rows, cin, cout, k)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += t
print("total %.2f ms in %d launches" % (tot / 1e3, n))
for key, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    rows, cin, cout, k = key
    print("rows=%7d cin=%4d cout=%4d K=%3d  x%2d  %8.1f us each  %8.1f us total" % (rows, cin, cout, k, cnt, t / cnt, t))
