"""What do capacity rows cost the planned forward?  The bench scene through pbnet_amd/planned.py at capacities of PROBE_SLACKS x its
measured sizes (kernel choice by the measured rows -- the rows hint -- in every case): ms per forward, one scene alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pbnet_amd import planned

dev = torch.device("cuda", 0)
cfg, model, b, t, info, raw = B.build_workload(0, 1, torch.bfloat16, dev, "c2", 1)
args = (b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"])
base = planned.measure_capacities(model, *args, teacher=t)
for slack in [float(v) for v in os.environ.get("PROBE_SLACKS", "1.0,1.25,2.0,4.0").split(",")]:
    pf = planned.PlannedForward(model, base.padded(slack), dtype=torch.bfloat16)
    for _ in range(3):
        pf(*args, teacher=t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = pf(*args, teacher=t)
    torch.cuda.synchronize()
    print("capacities %.2f x: %.3f ms per forward, %d proposals" % (slack, (time.perf_counter() - t0) / 20 * 1e3, out["proposals"][1].shape[0] - 1), flush=True)
