"""The serving front (pbnet_amd/serving.py) on the stream of distinct configs[1] scenes bench.py's `served` leg uses:
    served_probe.py B F [seconds=2]        -> scenes/s; run under `rocprofv3 --kernel-trace --stats` for the kernel time per scene
(scripts/kstats_summary.py counts forwards by k_centers calls: divide its per-forward figures by the scenes per forward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench
from pbnet_amd import synth
from pbnet_amd.serving import SceneServer

B, F = int(sys.argv[1]), int(sys.argv[2])
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
streams = bench.inflight_streams(dev, max(F, 4))
cfg, model = bench.build_model(dev)
scenes = []
for sd in range(2, 10):
    bt, tc, inf = synth.make_val_batch(copies=1, **dict(bench.WORKLOADS["c2"], seed=sd))
    sc = {k: torch.from_numpy(bt[k]).to(dev) for k in ("xyz_voxel", "feat_voxel", "xyz_original", "v2p_index")}
    sc["feat_voxel"] = sc["feat_voxel"].to(torch.bfloat16)
    scenes.append((sc, {k: torch.from_numpy(v).to(dev) for k, v in tc.items()}))
torch.cuda.synchronize()
srv = SceneServer(model, max_batch=B, forwards_in_flight=F, streams=streams[:F])


def burst(n):
    futs = [srv.submit(*scenes[i % 8]) for i in range(n)]
    return [f.result(timeout=600) for f in futs]


burst(2 * B * F)
torch.cuda.synchronize()
t0 = time.perf_counter(); burst(2 * B * F); dt = time.perf_counter() - t0
n = max(2 * B * F, int(2 * B * F * seconds / dt) // (B * F) * (B * F))
f0 = srv.forwards
torch.cuda.synchronize()
t0 = time.perf_counter(); burst(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("B%dxF%d: %.1f scenes/s (%d scenes in %d forwards, %.2f ms per scene)" % (B, F, n / dt, n, srv.forwards - f0, dt / n * 1e3))
srv.close()
