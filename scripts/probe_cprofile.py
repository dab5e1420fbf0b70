import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, cProfile, pstats
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(3): bench.one_step(model, b, t)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): bench.one_step(model, b, t)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
