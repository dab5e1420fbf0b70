import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pbnet_amd import prof
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, int(sys.argv[1]) if len(sys.argv) > 1 else 1, torch.bfloat16, dev)
for _ in range(3): bench.one_step(model, b, t)
prof.enable(); prof.reset()
for _ in range(5): bench.one_step(model, b, t)
tot = 0
for k, (ms, n) in prof.report().items():
    print("%-20s %8.3f ms" % (k, ms)); tot += ms
print("sum %.2f ms" % tot)
