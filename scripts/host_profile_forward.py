"""cProfile of the eager PBNet.forward's HOST side (one thread, inputs resident): where the ~3 ms of Python per forward go.
With four scenes in flight the interpreter lock is the shared resource (scripts/probe_threads.py: 4 x 3 ms of Python per 11.8 ms
step), so this is the in-flight rate's bound.  usage: host_profile_forward.py [steps] [sort: tottime|cumtime]"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sort = sys.argv[2] if len(sys.argv) > 2 else "tottime"
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(5):
    bench.one_step(model, b, t)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    bench.one_step(model, b, t)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats(sort)
ps.print_stats(45)
out = s.getvalue().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", "")
print("per forward: total %.3f ms" % (ps.total_tt / steps * 1e3))
print("\n".join(l for l in out.splitlines() if l.strip())[:9000])
