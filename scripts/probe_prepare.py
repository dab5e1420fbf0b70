"""pbn_coords_prepare on the bench scene's three lineages (backbone rows, local-scene rows, proposal rows): microseconds per
call of the sorted pipeline (csrc/pyramid.hip) and of the hash pipeline (pbn_coords_prepare_hash), back to back on one stream."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pbnet_amd import _native as N
from pbnet_amd.MinkowskiEngine import conventions as CV
import pbnet_amd.MinkowskiEngine.core as core

cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, torch.device("cuda", 0))
seen = []
orig = core.CoordinateManager.__init__
def rec(self, coordinates, *a, **k):
    seen.append(coordinates.to(torch.int32).contiguous().clone())
    return orig(self, coordinates, *a, **k)
core.CoordinateManager.__init__ = rec
bench.one_step(model, b, t)
core.CoordinateManager.__init__ = orig
lib = N.lib()
REP = 30
for coords in seen:
    n = int(coords.shape[0])
    P = N.PrepareLayout()
    nbytes = lib.pbn_coords_prepare_bytes(n, 1, ctypes.byref(P))
    arena = torch.empty(nbytes, dtype=torch.uint8, device=coords.device)
    res = []
    for name in ("sorted", "hash"):
        def call():
            if name == "sorted":
                rc = lib.pbn_coords_prepare(N.ptr(coords), n, 1, int(CV.X_FASTEST), N.ptr(arena), nbytes, ctypes.byref(P), N.current_stream())
            else:
                rc = lib.pbn_coords_prepare_hash(N.ptr(coords), None, n, 1, int(CV.X_FASTEST), N.ptr(arena), nbytes, ctypes.byref(P), N.current_stream())
            N.check(rc, name)
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP): call()
        e1.record(); torch.cuda.synchronize()
        res.append("%s %.1f us" % (name, e0.elapsed_time(e1) / REP * 1e3))
    print("rows %7d: %s" % (n, "  ".join(res)), flush=True)

if os.environ.get("PBN_SORT_STAMPS"):
    coords = seen[0]; n = int(coords.shape[0])
    P = N.PrepareLayout(); nbytes = lib.pbn_coords_prepare_bytes(n, 1, ctypes.byref(P))
    arena = torch.empty(nbytes, dtype=torch.uint8, device=coords.device)
    for _ in range(3):
        lib.pbn_coords_prepare(N.ptr(coords), n, 1, int(CV.X_FASTEST), N.ptr(arena), nbytes, ctypes.byref(P), N.current_stream())
    torch.cuda.synchronize()
    plan = arena[P.sort_temp:P.sort_temp + 256].view(torch.int32).cpu().numpy()
    import numpy as np
    for nm, o in (("block 0", 16), ("block 20", 32)):
        st = plan[o:o + 8].astype(np.int64) & 0xffffffff
        print(nm, "stamps (cycles from start):", [int((x - st[0]) & 0xffffffff) for x in st])
