import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pbnet_amd import prof
from pbnet_amd.MinkowskiEngine import conv as C
from pbnet_amd.network import mink_unet as U
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(3): bench.one_step(model, b, t)
recs = []
orig = C.spconv_forward
def wrapped(feats, nbr, n_out, packed, **kw):
    t0 = time.perf_counter(); out = orig(feats, nbr, n_out, packed, **kw); t1 = time.perf_counter()
    recs.append((t1 - t0, int(n_out), int(feats.shape[1]), packed[3], 1 if nbr is None else nbr.shape[1]))
    return out
C.spconv_forward = wrapped; U.spconv_forward = wrapped
prof.enable(); prof.reset()
st0 = torch.cuda.memory_stats()
for _ in range(3): bench.one_step(model, b, t)
st1 = torch.cuda.memory_stats()
print({k: st1[k] - st0[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries") if k in st1})
recs.sort(reverse=True)
print("slowest CPU-side launches (s, rows, cin, cout, K):")
for r in recs[:10]: print("  %.6f" % r[0], r[1:])
print("total CPU in spconv_forward per step: %.2f ms" % (sum(r[0] for r in recs) / 3 * 1e3))
for k, (ms, n) in prof.report().items(): print("%-20s %8.3f ms" % (k, ms))
