"""Where the host threads of an in-flight run spend their time: waiting for the device (read-backs), inside native calls
(launch sequences, GIL released) or in Python (GIL held, includes waiting for the GIL)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
from pbnet_amd import _native as N
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = 40
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(3): bench.one_step(model, b, t)
torch.cuda.synchronize()
tl = threading.local()
def acc(kind, dt):
    d = getattr(tl, "d", None)
    if d is not None: d[kind] = d.get(kind, 0.0) + dt
def timed(fn, kind):
    def w(*a, **k):
        t0 = time.perf_counter()
        try: return fn(*a, **k)
        finally: acc(kind, time.perf_counter() - t0)
    return w
for name in ("cpu", "tolist", "item"):
    setattr(torch.Tensor, name, timed(getattr(torch.Tensor, name), "sync"))
lib = N.lib()
class Proxy(object):
    def __getattr__(self, name):
        f = getattr(lib, name)
        w = timed(f, "native") if callable(f) else f
        setattr(self, name, w)
        return w
N._lib = Proxy()
res = {}
def worker(i, n):
    torch.cuda.set_device(dev)
    tl.d = {}
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        bench.one_step(model, b, t); s.synchronize()
        tl.d = {}
        t0 = time.perf_counter()
        for _ in range(n): bench.one_step(model, b, t)
        s.synchronize()
        tl.d["wall"] = time.perf_counter() - t0
    res[i] = tl.d
ths = [threading.Thread(target=worker, args=(i, STEPS)) for i in range(M)]
T0 = time.perf_counter()
for th in ths: th.start()
for th in ths: th.join()
T = time.perf_counter() - T0
print("%d in flight: %.1f scenes/s" % (M, M * STEPS / T))
for i in sorted(res):
    d = res[i]; w = d["wall"] / STEPS * 1e3
    s, n = d.get("sync", 0) / STEPS * 1e3, d.get("native", 0) / STEPS * 1e3
    print("thread %d: step %.2f ms = device wait %.2f + native calls %.2f + python %.2f" % (i, w, s, n, w - s - n))
