"""cProfile of the sync-free (capacity-planned) forward's HOST side, one thread: Python per forward when nothing waits for the
device.  usage: host_profile_planned.py [steps]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pbnet_amd import planned

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
dtype = torch.bfloat16
cfg, model, b, t, info, raw = bench.build_workload(0, 1, dtype, dev)
args = (b["feat_voxel"].to(dtype), b["xyz_voxel"], b["xyz_original"], b["v2p_index"])
cap = planned.measure_capacities(model, *args, teacher=t).padded(1.25)
pf = planned.PlannedForward(model, cap, dtype=dtype)
for _ in range(5):
    pf(*args, teacher=t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pf(*args, teacher=t)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("planned forward: host issue %.3f ms per forward, with the device drained %.3f ms" % (t_issue / steps * 1e3, t_all / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    pf(*args, teacher=t)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(30)
out = s.getvalue().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", "")
print("per forward (profiled): %.3f ms" % (ps.total_tt / steps * 1e3))
print("\n".join(l for l in out.splitlines() if l.strip())[:6000])
