"""Single-layer conv timing (S150 stride-1 level, 96->96, K=27, bf16) under PBN_CONV_DBG ablations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward
dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
coords = torch.from_numpy(batch["xyz_voxel"]).to(dev)
cm = ME.CoordinateManager(coords)
SORTED = os.environ.get("PBN_PROBE_SORTED", "0") == "1"
pyr = cm.sorted().pyramid if SORTED else cm.plain()
torch.manual_seed(0)
def run(level, cin, cout, k=3, rw=0):
    stride = 1 << level
    n = pyr.n[level]
    nbr = pyr.kernel_map(stride, k)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3).to(dev)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    packed = conv._cache.get(conv.kernel, torch.bfloat16)
    for _ in range(3): spconv_forward(x, nbr, n, packed, rows_per_wave=rw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): spconv_forward(x, nbr, n, packed, rows_per_wave=rw)
    e1.record(); torch.cuda.synchronize()
    pairs = int((nbr >= 0).sum().item())
    # fragment-level fill: fraction of (16-row fragment, offset) pairs with at least one neighbour
    nb16 = (nbr[: n // 16 * 16].view(-1, 16, nbr.shape[1]) >= 0).any(1).float().mean().item()
    nb128 = (nbr[: n // 128 * 128].view(-1, 128, nbr.shape[1]) >= 0).any(1).float().mean().item()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print("dbg=%s level=%d rows=%d %d->%d K=%d rw=%d: %.1f us  (pairs/row %.2f, fill16 %.2f, fill128 %.2f, %.1f TFLOP/s real) sorted=%s" % (
        os.environ.get("PBN_CONV_DBG", "0"), level, n, cin, cout, k ** 3, rw, t, pairs / n, nb16, nb128, 2 * pairs * cin * cout / t / 1e6, SORTED))
RWS = [int(v) for v in os.environ.get("PBN_PROBE_RWS", "16,32").split(",")]
CASES = {"0": (0, 96, 96), "1": (1, 96, 96), "2": (2, 128, 128), "3": (3, 256, 256), "3n": (3, 128, 128), "4": (4, 256, 256)}
for rw in RWS:
    for c in os.environ.get("PBN_PROBE_CASES", "0,1,2,3,3n,4").split(","):
        run(*CASES[c], rw=rw)
