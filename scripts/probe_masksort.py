"""Experiment: rows of a level re-ordered so that rows with the same neighbour mask share 16-row fragments (sorted by the
27-bit population mask inside blocks of B consecutive Z-order rows), through pbn_spconv_forward's row_perm argument.
Prints microseconds per launch (HIP-graph replay) against the natural order, and checks the outputs are identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward
dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
REP = 20


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


def run(level, cin, cout):
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, 3)
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3).to(dev)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    packed = conv._cache.get(conv.kernel, torch.bfloat16)
    out0 = torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev)
    t0 = timed(lambda: spconv_forward(x, nbr, n, packed, rows_per_wave=32, out=out0))
    pop = nbr >= 0
    bits = (pop.long() << torch.arange(27, device=dev)).sum(1)
    line = "L%d rows=%6d %3d->%3d: natural %.1f us" % (level, n, cin, cout, t0)
    for B in (128, 512, 2048, 8192):
        blk = torch.arange(n, device=dev) // B
        key = blk * (1 << 27) + bits
        perm = torch.argsort(key, stable=True).to(torch.int32)
        out1 = torch.zeros_like(out0)
        t1 = timed(lambda: spconv_forward(x, nbr, n, packed, rows_per_wave=32, out=out1, row_perm=perm))
        # same rows through the permuted table with the fast prologue (nbr pre-permuted, outputs in tile order)
        nbr_p = nbr[perm.long()].contiguous()
        out2 = torch.zeros_like(out0)
        t2 = timed(lambda: spconv_forward(x, nbr_p, n, packed, rows_per_wave=32, out=out2))
        ok = torch.equal(out1, out0) and torch.equal(out2, out0[perm.long()])
        frag = pop[perm.long()][: n // 16 * 16].reshape(-1, 16, 27).any(1).float().mean().item()
        tile = pop[perm.long()][: n // 128 * 128].reshape(-1, 128, 27).any(1).float().mean().item()
        line += " | B=%d: %.1f (pre-permuted table %.1f) frag16 %.2f tile128 %.2f %s" % (B, t1, t2, frag, tile, "ok" if ok else "MISMATCH")
    print(line, flush=True)


for c in [(0, 96, 96), (1, 96, 96), (1, 32, 32), (2, 128, 128), (2, 64, 64)]:
    run(*c)
