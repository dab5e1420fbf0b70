"""Single-layer conv timing with the host taken out: REP launches captured in a HIP graph, replayed, HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward
dev = "cuda:0"
seed = int(os.environ.get("PBN_PROBE_SEED", "2"))
batch, _, _ = synth.make_val_batch(seed=seed, copies=1)
coords = torch.from_numpy(batch["xyz_voxel"]).to(dev)
cm = ME.CoordinateManager(coords)
pyr = cm.sorted().pyramid
torch.manual_seed(0)
REP = 20
def run(level, cin, cout, k=3, rw=0):
    stride = 1 << level
    n = pyr.n[level]
    nbr = pyr.kernel_map(stride, k)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3).to(dev)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    packed = conv._cache.get(conv.kernel, torch.bfloat16)
    out = torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev)
    for _ in range(3): spconv_forward(x, nbr, n, packed, rows_per_wave=rw, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): spconv_forward(x, nbr, n, packed, rows_per_wave=rw, out=out)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (2 * REP) * 1e3
    print("dbg=%s split=%s L%d rows=%d %d->%d K=%d rw=%d: %.1f us" % (os.environ.get("PBN_CONV_DBG", "0"),
          os.environ.get("PBN_CONV_SPLIT", "-"), level, n, cin, cout, k ** 3, rw, t))
CASES = {"0": (0, 96, 96), "1": (1, 96, 96), "2": (2, 128, 128), "2n": (2, 64, 64), "3": (3, 256, 256), "3n": (3, 128, 128), "4": (4, 256, 256)}
for rw in [int(v) for v in os.environ.get("PBN_PROBE_RWS", "16,32").split(",")]:
    for c in os.environ.get("PBN_PROBE_CASES", "0,1,2,3,3n,4").split(","):
        run(*CASES[c], rw=rw)
