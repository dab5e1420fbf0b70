import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from pbnet_amd import pbnet_ops
from test_cluster_gpu import _scene_groups
dev = "cuda:0"
for copies in (1, 3):
    off, org, sem, seg = _scene_groups(seed=2, pitch=0.0225, room=(4.0, 3.2, 2.6), n_boxes=12, copies=copies)
    t = lambda a: torch.from_numpy(a).to(dev)
    o, g, s, l = t(off), t(org), t(sem), t(seg)
    for _ in range(3): pbnet_ops.cluster_device(o, g, s, l, 0.04, 31)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pbnet_ops.cluster_device(o, g, s, l, 0.04, 31)
    torch.cuda.synchronize()
    print("grouping %d points %d segments: %.3f ms" % (len(off), len(seg), (time.perf_counter() - t0) / 20 * 1e3))
