import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pbnet_amd.network import mink_unet as U
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(3): bench.one_step(model, b, t)
log = []
orig_fused = U.MinkUNet._forward_fused
def timed_fused(self, x):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig_fused(self, x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    log.append((self.arch, x.F.shape[0], (t1 - t0) * 1e3, (t2 - t0) * 1e3))
    return out
U.MinkUNet._forward_fused = timed_fused
for _ in range(3): bench.one_step(model, b, t)
for r in log: print("%s rows=%d  cpu_launch=%.2f ms  total=%.2f ms" % r)
