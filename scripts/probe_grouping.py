"""The grouping stage (pbn_binary_cluster) on the bench scene's selected points: microseconds per call from a HIP graph
(all of its launches, one scene alone on the GPU) -- for A/B of library builds (scripts/ab_libs.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
from pbnet_amd import pbnet_ops

dev = torch.device("cuda", 0)
cfg, model, b, t, info, raw = B.build_workload(0, 1, torch.bfloat16, dev, "c2", 1)
calls = []
orig = pbnet_ops.cluster_device


def spy(*a, **k):
    calls.append((a, k))
    return orig(*a, **k)


pbnet_ops.cluster_device = spy
import pbnet_amd.network.PBNet as P
if hasattr(P, "cluster_device"):
    P.cluster_device = spy
B.one_step(model, b, t)
torch.cuda.synchronize()
assert calls, "the forward did not reach the grouping"
a, k = calls[0]
pbnet_ops.cluster_device = orig
for _ in range(3):
    r = orig(*a, **k)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        r = orig(*a, **k)
g.replay(); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10 * 1e6)
c = r.centers[:3 * int(r.n_clusters.item())].cpu().numpy()
print("grouping of %d points, %d clusters: %.1f us per call; centers checksum %s" % (a[0].shape[0], int(r.n_clusters.item()), best, hex(int(np.frombuffer(c.tobytes(), dtype=np.uint32).astype(np.uint64).sum()))))

# capacity mode (the planned forward, the device front of the size-exact forward): the same points in buffers of `cap` rows, the
# count on the device -- what do the extra rows cost?   PROBE_CAPS="1.0,2.5,7.5" (multiples of the point count)
caps = [float(v) for v in os.environ.get("PROBE_CAPS", "").split(",") if v]
for mult in caps:
    off, org, sem, seg_len = a[0], a[1], a[2], a[3]
    m = off.shape[0]
    cap = int(m * mult)
    pad = lambda t: torch.cat([t, torch.zeros((cap - m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)]) if cap > m else t
    args = (pad(off), pad(org), pad(sem), seg_len) + tuple(a[4:])
    kw = dict(k, capacity=True)
    for _ in range(3):
        r2 = orig(*args, **kw)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for _ in range(10):
            r2 = orig(*args, **kw)
    g2.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); g2.replay(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10 * 1e6)
    same = torch.equal(r2.cluster_id[:m], r.cluster_id) and int(r2.n_clusters.item()) == int(r.n_clusters.item())
    print("capacity %.1f x (%d rows): %.1f us per call, same clusters: %s" % (mult, cap, best, same))
