"""PBNet.forward with its capacity cache on / off and the explicit planned forward, alternating, on the bench scene with
--inflight host threads: scenes/s of each leg per round (same process, same box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench as B

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg, model, b, t, info, raw = B.build_workload(0, 1, torch.bfloat16, dev, "c2", 1)
INF = int(os.environ.get("AB_INFLIGHT", "4"))
K = int(os.environ.get("AB_STEPS", "60"))


def leg(on):
    model.planned_cache = on
    r = B.Runner(model, b, t, INF, dev)
    r.run(2 * INF)
    r.run(2 * INF)
    bl = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r.run(K); torch.cuda.synchronize(); bl.append(time.perf_counter() - t0)
    return K / float(np.median(bl))


for rnd in range(int(os.environ.get("AB_ROUNDS", "3"))):
    off = leg(False)
    on = leg(True)
    path = model.forward_path()
    model.planned_cache = False
    pl = B.planned_leg(model, b, t, torch.bfloat16, INF, dev, K, graph=False)["value"]
    print("round %d: size-exact %.1f  cached %.1f  explicit planned %.1f scenes/s" % (rnd, off, on, pl), flush=True)
