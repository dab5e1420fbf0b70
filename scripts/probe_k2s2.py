"""The k=2,s=2 convolutions of the bench scene (strided down, transposed up), every kernel configuration, HIP-graph replay."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward, pack_weight
dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
REP = 20
CFGS = [int(c) for c in os.environ.get("PBN_PROBE_CFGS", "0,16,32,1201,1202,1204,1401,1402,1404").split(",")]


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


def run(kind, level, cin, cout):
    # down: level = fine level, out rows = n[level + 1]; up: level = fine level (output), in rows = n[level + 1]
    if kind == "down":
        nbr, n_in, n_out = pyr.down_map(1 << level), pyr.n[level], pyr.n[level + 1]
    else:
        nbr, n_in, n_out = pyr.up_map(2 << level), pyr.n[level + 1], pyr.n[level]
    torch.manual_seed(0)
    w = torch.randn(8, cin, cout, device=dev) * 0.05
    packed = pack_weight(w, torch.bfloat16)
    x = torch.randn(n_in, cin, device=dev).to(torch.bfloat16)
    out = torch.empty(n_out, packed[3], dtype=torch.bfloat16, device=dev)
    res = []
    for cfg in CFGS:
        if cfg >= 100 and (packed[3] // 16) % (cfg % 100):
            continue
        try:
            spconv_forward(x, nbr, n_out, packed, rows_per_wave=cfg, out=out)
        except RuntimeError:
            continue
        res.append((cfg, timed(lambda: spconv_forward(x, nbr, n_out, packed, rows_per_wave=cfg, out=out))))
    pop = (nbr >= 0).float().mean().item()
    print("%-4s L%d in %6d out %6d %3d->%3d (table %.2f populated): " % (kind, level, n_in, n_out, cin, cout, pop) +
          "  ".join("%d:%.1f" % r for r in res), flush=True)


for c in (("down", 0, 32, 32), ("down", 1, 32, 64), ("down", 2, 64, 128), ("down", 3, 128, 256),
          ("up", 3, 256, 256), ("up", 2, 256, 128), ("up", 1, 128, 96), ("up", 0, 96, 96)):
    run(*c)
