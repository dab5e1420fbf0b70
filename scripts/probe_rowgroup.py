"""Does regrouping the rows of a 128-row tile by neighbour mask pay on the wide layers?  (round 4 experiment)
Rows of a tile are re-ordered so that the 16 rows of a fragment have similar masks (greedy: seed = densest unassigned row, then the
rows that grow the union least); the convolution runs with row_perm = that order.  Prints the issued (fragment, offset) share
before / after and microseconds per launch: no perm / identity perm (the price of the permuted prologue) / greedy perm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward

dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
REP = 20
DT = torch.bfloat16


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * REP) * 1e3


def popc(x):
    c = np.zeros_like(x)
    for i in range(27):
        c += (x >> i) & 1
    return c


def greedy_perm(bits, tile=128, frag=16):
    n = len(bits)
    nt = n // tile
    b = bits[:nt * tile].reshape(nt, tile)
    pc = popc(b)
    assigned = np.zeros((nt, tile), dtype=bool)
    ar = np.arange(nt)
    order = np.zeros((nt, tile), dtype=np.int64)
    pos = 0
    for g in range(tile // frag):
        s = np.where(assigned, -1, pc).argmax(1)
        union = b[ar, s].copy(); assigned[ar, s] = True; order[:, pos] = s; pos += 1
        for j in range(frag - 1):
            grow = popc(b | union[:, None]) - popc(union)[:, None]
            s = np.where(assigned, 1 << 20, grow * 64 - pc).argmin(1)
            union |= b[ar, s]; assigned[ar, s] = True; order[:, pos] = s; pos += 1
    perm = np.arange(n, dtype=np.int64)
    perm[:nt * tile] = (order + (ar * tile)[:, None]).reshape(-1)
    return perm


def issued(bits, perm, frag=16):
    b = bits[perm]
    m = len(b) // frag * frag
    u = np.bitwise_or.reduce(b[:m].reshape(-1, frag), axis=1)
    return popc(u).sum() * frag / (m * 27)


def run(level, cin, cout):
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, 3).contiguous()
    h = nbr.cpu().numpy()
    bits = ((h >= 0).astype(np.int64) << np.arange(27)).sum(1)
    t0 = time.perf_counter()
    perm = greedy_perm(bits)
    ident = np.arange(n, dtype=np.int64)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3).to(dev)
    packed = conv._cache.get(conv.kernel, DT)
    x = torch.randn(n, packed[1] * 8, device=dev).to(DT)
    out = torch.empty(n, packed[3], dtype=DT, device=dev)
    o1, o2 = torch.empty_like(out), torch.empty_like(out)
    p_i = torch.from_numpy(ident.astype(np.int32)).to(dev)
    p_g = torch.from_numpy(perm.astype(np.int32)).to(dev)
    t_none = timed(lambda: spconv_forward(x, nbr, n, packed, out=out))
    t_id = timed(lambda: spconv_forward(x, nbr, n, packed, out=o1, row_perm=p_i))
    t_g = timed(lambda: spconv_forward(x, nbr, n, packed, out=o2, row_perm=p_g))
    err = max((out.float() - o1.float()).abs().max().item(), (out.float() - o2.float()).abs().max().item())
    print("L%d rows=%6d %3d->%3d: issued %.3f -> %.3f (populated %.3f) | no perm %.1f us, identity perm %.1f, greedy perm %.1f  (max diff %.1e)" % (
        level, n, cin, cout, issued(bits, ident), issued(bits, perm), popc(bits).sum() / (n * 27), t_none, t_id, t_g, err), flush=True)


cases = [(0, 96, 96), (0, 128, 96), (1, 96, 96), (1, 128, 96), (1, 32, 32), (2, 64, 64), (2, 128, 128)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for c in cases:
    run(*c)
