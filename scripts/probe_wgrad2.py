"""pbn_spconv_wgrad layer by layer on the bench scene under the measurement knobs of csrc/wgrad.hip (PBN_WGRAD_WGS = workgroup
target -> pair splits, PBN_WGRAD_DBG 1 = no main loop / 2 = no stores): microseconds per call (HIP events, 20 calls)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth, _native as N
from pbnet_amd.MinkowskiEngine import conv as C
dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


DBGS = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2".split(","))]
WGS = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,256,512,1024,2048".split(","))]
for level, cin, cout in ((4, 256, 256), (3, 256, 256), (3, 128, 128), (2, 128, 128), (1, 96, 96), (0, 128, 96), (0, 96, 96), (1, 32, 32)):
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, 3)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    g = torch.randn(n, cout, device=dev).to(torch.bfloat16)
    pairs = int((nbr >= 0).sum())
    line = "L%d %3d->%3d rows %6d pairs %7d:" % (level, cin, cout, n, pairs)
    for w in WGS:
        os.environ["PBN_WGRAD_WGS"] = str(abs(w))
        os.environ["PBN_WGRAD_MAXT"] = os.environ.get("MAXT", "4")
        cells = []
        for dbg in DBGS:
            os.environ["PBN_WGRAD_DBG"] = str(dbg)
            cells.append("%.0f" % timed(lambda: C.wgrad_native(x, g, nbr, cin, cout)))
        line += "  wgs%d %s" % (w, "/".join(cells))
    os.environ["PBN_WGRAD_DBG"] = "0"
    print(line, flush=True)
