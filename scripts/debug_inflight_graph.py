"""Throughput probe: M host threads, each with its own captured planned forward (HIP graph) on its own stream."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from pbnet_amd import planned, synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet
DEV = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = sys.argv[2] if len(sys.argv) > 2 else "graph"
STEPS = 60
cfg = get_config(test=True)
torch.manual_seed(22)
model = PBNet(cfg).to(DEV).eval()
batch, teacher, info = synth.make_val_batch(seed=2, copies=1)
b = {k: torch.from_numpy(v).to(DEV) for k, v in batch.items()}
b["feat_voxel"] = b["feat_voxel"].to(torch.bfloat16)
t = {k: torch.from_numpy(v).to(DEV) for k, v in teacher.items()}
args = (b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"])
cap = planned.measure_capacities(model, *args, teacher=t).padded(1.25)
streams = [torch.cuda.Stream(DEV) for _ in range(M)]
pfs = []
for i in range(M):
    with torch.cuda.stream(streams[i]):
        pf = planned.PlannedForward(model, cap, dtype=torch.bfloat16)
        pf(*args, teacher=t)
        if MODE in ("graph", "graph_nosync"):
            pf.capture(*args, teacher=t)
        pfs.append(pf)
    torch.cuda.synchronize()
print("captured %d" % M, flush=True)

def worker(i, n):
    torch.cuda.set_device(DEV)
    with torch.cuda.stream(streams[i]):
        for _ in range(i, n, M):
            if MODE == "graph_nosync":          # no read-back at all between replays: the device-side ceiling
                pfs[i].replay()
            elif MODE == "graph":
                pfs[i].finish(pfs[i].replay())
            else:
                pfs[i](*args, teacher=t)
        streams[i].synchronize()

def run(n):
    th = [threading.Thread(target=worker, args=(i, n)) for i in range(M)]
    for x in th: x.start()
    for x in th: x.join()
run(2 * M)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(STEPS)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%s, %d in flight: %.1f scenes/s (%.3f ms/scene)" % (MODE, M, STEPS / el, el / STEPS * 1e3), flush=True)
