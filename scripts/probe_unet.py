"""The three U-Net executor calls of one bench step (MinkUNet34C backbone, MinkUNet14A mask, MinkUNet34C score), each replayed
alone from a HIP graph (no host time, no events between ops): ms per U-Net forward and microseconds per convolution op.
The inputs are the real ones of the bench scene (captured from one forward)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pbnet_amd.network import mink_unet as U

cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, torch.device("cuda", 0))
for _ in range(2):
    bench.one_step(model, b, t)
calls = []
orig = U.MinkUNet._forward_fused


def rec(self, x):
    calls.append((self, x))
    return orig(self, x)


U.MinkUNet._forward_fused = rec
bench.one_step(model, b, t)
U.MinkUNet._forward_fused = orig
torch.cuda.synchronize()
REP = 10
tot = 0.0
for net, x in calls:
    for _ in range(2):
        orig(net, x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            out = orig(net, x)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * REP)
    n_ops = net._plan(x.F.dtype)["n_ops"]
    rows = x.coordinate_manager.sorted().pyramid.n
    print("%s rows %s: %.3f ms per forward, %d ops, %.1f us per op" % (net.arch, rows, ms, n_ops, ms * 1e3 / n_ops), flush=True)
    tot += ms
print("three U-Nets: %.3f ms" % tot)
