"""Host-side timeline of one step: where the Python thread spends its time (no extra device syncs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pbnet_amd.network import mink_unet as U
from pbnet_amd.MinkowskiEngine import core as C
from pbnet_amd import _native as N

cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, torch.device("cuda", 0))
for _ in range(5):
    bench.one_step(model, b, t)
from pbnet_amd import prof
marks = prof.host_marks(True)
def mark(name):
    prof.mark(name)

orig_fin = C.CoordinateManager._finalize
def fin(self):
    if self._final:
        return orig_fin(self)
    mark("finalize:enter")
    r = orig_fin(self)
    mark("finalize:exit")
    return r
C.CoordinateManager._finalize = fin
orig_ff = U.MinkUNet._forward_fused
def ff(self, x):
    mark("fused:enter")
    r = orig_ff(self, x)
    mark("fused:exit")
    return r
U.MinkUNet._forward_fused = ff
lib = N.lib()
orig_tolist = torch.Tensor.tolist
def tl(self):
    mark("tolist:enter")
    r = orig_tolist(self)
    mark("tolist:exit")
    return r
torch.Tensor.tolist = tl
orig_cpu = torch.Tensor.cpu
def cpu(self, *a, **k):
    mark("cpu:enter")
    r = orig_cpu(self, *a, **k)
    mark("cpu:exit")
    return r
torch.Tensor.cpu = cpu
torch.cuda.synchronize()
for rep in range(3):
    del marks[:]
    t0 = time.perf_counter()
    mark("step:start")
    bench.one_step(model, b, t)
    mark("step:launched")
    torch.cuda.synchronize()
    mark("step:done")
prev = marks[0][1]
for name, tm in marks:
    print("%-18s +%7.1f us   @%8.1f us" % (name, (tm - prev) * 1e6, (tm - marks[0][1]) * 1e6))
    prev = tm
