"""What would ordering rows by (spatial block, neighbour-mask) buy the big stride-1 layers?  Permute the level's table
in Python and time the unchanged kernel (HIP-graph replay, no host time)."""
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
pyr = cm.sorted().pyramid
REP = 20
def timed(x, nbr, n, packed):
    out = torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev)
    for _ in range(3): spconv_forward(x, nbr, n, packed, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): spconv_forward(x, nbr, n, packed, out=out)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * REP) * 1e3, out
def fill(nbr, T):
    n = nbr.shape[0] // T * T
    return (nbr[:n].view(-1, T, nbr.shape[1]) >= 0).any(1).float().mean().item()
def run(level, cin, cout):
    stride = 1 << level
    n = pyr.n[level]
    nbr = pyr.kernel_map(stride, 3)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3).to(dev)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    packed = conv._cache.get(conv.kernel, torch.bfloat16)
    t0, o0 = timed(x, nbr, n, packed)
    print("L%d rows %d: Z-order            %.1f us  fill16 %.2f fill128 %.2f" % (level, n, t0, fill(nbr, 16), fill(nbr, 128)))
    valid = nbr >= 0
    w = (1 << torch.arange(27, device=dev, dtype=torch.int64))
    mask = (valid.long() * w).sum(1)
    for blk in (0, 512, 2048, 8192, 32768):
        pos = torch.arange(n, device=dev)
        key = mask if blk == 0 else (pos // blk) * (1 << 27) + mask
        p = torch.sort(key, stable=True)[1]
        inv = torch.empty_like(p); inv[p] = torch.arange(n, device=dev)
        nb2 = nbr[p].long()
        nb2 = torch.where(nb2 >= 0, inv[nb2.clamp(min=0)], nb2).to(torch.int32).contiguous()
        t1, o1 = timed(x[p].contiguous(), nb2, n, packed)
        err = (o1.float() - o0[p].float()).abs().max().item()
        print("   block %6d + mask sort: %.1f us  fill16 %.2f fill128 %.2f  (max diff %.3g)" % (blk, t1, fill(nb2, 16), fill(nb2, 128), err))
run(0, 96, 96)
run(1, 96, 96)
run(2, 128, 128)
