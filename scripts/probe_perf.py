"""Ad-hoc timing probe (not the bench): U-Net forward and grouping on the S150 synthetic scene."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth, pbnet_ops
from pbnet_amd.network.Mink import Mink_unet

dev = "cuda:0"
sc = synth.synth_room(seed=2, pitch=0.0225, room=(4.0, 3.2, 2.6), n_boxes=12)
q, first, inv = synth.voxelize_numpy(sc["xyz"], 0.02)
coords = torch.from_numpy(np.concatenate([np.zeros((len(q), 1), np.int32), q], 1).astype(np.int32)).to(dev)
print("points", len(sc["xyz"]), "voxels", len(q))

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

torch.manual_seed(22)
net = Mink_unet(6, 32, arch="MinkUNet34C").to(dev).eval()
for dt in (torch.float32, torch.bfloat16):
    feats = torch.randn(len(q), 6, device=dev).to(dt)
    def fwd():
        with torch.no_grad():
            x = ME.SparseTensor(feats, coords)
            return net(x).F
    print("34C forward incl. coordinate build", dt, "%.2f ms" % timeit(fwd))
    x = ME.SparseTensor(feats, coords)
    with torch.no_grad(): net(x)
    def fwd2():
        with torch.no_grad(): return net(x).F
    print("34C forward, maps cached        ", dt, "%.2f ms" % timeit(fwd2))
def cm_only():
    cm = ME.CoordinateManager(coords); cm.level(16); cm.kernel_map(1,5); [cm.kernel_map(s,3) for s in (1,2,4,8,16)]; [cm.up_map(s) for s in (2,4,8,16)]
print("coordinate pyramid + all maps    %.2f ms" % timeit(cm_only))

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_cluster_gpu import _scene_groups
off, org, sem, seg = _scene_groups(seed=2, pitch=0.0225, room=(4.0, 3.2, 2.6), n_boxes=12, copies=3)
t = lambda a: torch.from_numpy(a).to(dev)
o, g, s, l = t(off), t(org), t(sem), t(seg)
def grp(): return pbnet_ops.cluster_device(o, g, s, l, 0.04, 31)
print("grouping %d points, %d segments   %.2f ms" % (len(off), len(seg), timeit(grp)))
