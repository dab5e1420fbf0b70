import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
mode = sys.argv[1]
if mode == "threads": torch.set_num_threads(8)
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(3): bench.one_step(model, b, t)
torch.cuda.synchronize()
if mode == "nogc": gc.disable()
if mode == "freeze": gc.collect(); gc.freeze()
ts = []
for i in range(24):
    t0 = time.perf_counter(); bench.one_step(model, b, t); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(mode, "threads", torch.get_num_threads(), " ".join("%.0f" % x for x in ts))
