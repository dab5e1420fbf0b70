"""One convolution layer of the bench scene per (level, channels) case, every kernel configuration, replayed from a HIP
graph (no host time): microseconds per launch.  rows_per_wave 32 = workgroup-tile family (spconv.hip, incl. its split-K
reduce launch); >= 100 = wave family configurations (spconv_wave.hip)."""
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
torch.manual_seed(0)
REP = 20
STREAMS = int(os.environ.get("PBN_PROBE_STREAMS", "1"))      # > 1: the same layer on that many streams at once (throughput mode)
CFGS = [int(c) for c in os.environ.get("PBN_PROBE_CFGS", "32,401,402,404,406,408,204,206,208,1401,1402,1404,1201,1202,1204").split(",")]


def run(level, cin, cout, k=3):
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, k) if k > 1 else None
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3).to(dev)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    packed = conv._cache.get(conv.kernel, torch.bfloat16)
    out = torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev)
    res = []
    for cfg in CFGS:
        if cfg >= 100 and (packed[3] // 16) % (cfg % 100):
            continue
        try:
            for _ in range(2):
                spconv_forward(x, nbr, n, packed, rows_per_wave=cfg, out=out)
        except RuntimeError:
            continue
        torch.cuda.synchronize()
        if STREAMS > 1:
            streams = [torch.cuda.Stream() for _ in range(STREAMS)]
            outs = [torch.empty_like(out) for _ in range(STREAMS)]
            graphs = []
            for st, o in zip(streams, outs):
                with torch.cuda.stream(st):
                    for _ in range(2):
                        spconv_forward(x, nbr, n, packed, rows_per_wave=cfg, out=o)
                torch.cuda.synchronize()
                gq = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gq, stream=st):
                    for _ in range(REP):
                        spconv_forward(x, nbr, n, packed, rows_per_wave=cfg, out=o)
                graphs.append(gq)
            def go():
                for st, gq in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        gq.replay()
            go(); torch.cuda.synchronize()
            import time
            t0 = time.perf_counter(); go(); go(); torch.cuda.synchronize(); t1 = time.perf_counter()
            res.append((cfg, (t1 - t0) / (2 * REP * STREAMS) * 1e6))
            continue
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(REP):
                spconv_forward(x, nbr, n, packed, rows_per_wave=cfg, out=out)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        res.append((cfg, e0.elapsed_time(e1) / (2 * REP) * 1e3))
    best = min(res, key=lambda r: r[1])
    print("L%d rows=%6d %3d->%3d K=%3d: " % (level, n, cin, cout, k ** 3) + "  ".join("%d:%.1f" % r for r in res) + "   best %d" % best[0], flush=True)


CASES = [(4, 256, 256, 3), (4, 128, 256, 1), (3, 128, 128, 3), (3, 256, 256, 3), (3, 384, 256, 3), (2, 64, 64, 3), (2, 128, 128, 3),
         (2, 32, 64, 1), (1, 32, 32, 3), (1, 96, 96, 3), (0, 96, 96, 3), (0, 8, 32, 5)]
sel = os.environ.get("PBN_PROBE_CASES")
for i, c in enumerate(CASES):
    if sel is None or str(i) in sel.split(","):
        run(*c)
