import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pbnet_amd import prof
import pbnet_amd.network.PBNet as PB
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(3): bench.one_step(model, b, t)
# monkeypatch section() to record CPU timestamps without syncing
marks = []
import contextlib
@contextlib.contextmanager
def section(name):
    t0 = time.perf_counter()
    yield
    marks.append((name, t0, time.perf_counter()))
PB.section = section
import pbnet_amd.network.mink_unet as MU
MU.section = section
torch.cuda.synchronize()
T0 = time.perf_counter()
NS = 7
for _ in range(NS):
    marks.append(("STEP", time.perf_counter(), time.perf_counter()))
    bench.one_step(model, b, t)
    torch.cuda.synchronize()
T1 = time.perf_counter()
print("avg step %.2f ms" % ((T1 - T0) / NS * 1e3))
for name, a, e in marks:
    if name == "STEP" or (e - a) > 2e-3:
        print("%-20s start=%8.3f ms dur=%8.3f ms" % (name, (a - T0) * 1e3, (e - a) * 1e3))
