"""The LDS-staged family (spconv_halo.hip) against the gather kernels on layers of the bench scene, replayed from a HIP graph:
microseconds per launch, plus the table build.  PBN_PROBE_STREAMS=4: the same layer on four streams at once."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward, spconv_forward_halo, HaloTable
dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
torch.manual_seed(0)
REP = 20
STREAMS = int(os.environ.get("PBN_PROBE_STREAMS", "1"))
SLOTS = [int(s) for s in os.environ.get("PBN_PROBE_SLOTS", "0").split(",")]
TMS = [int(s) for s in os.environ.get("PBN_PROBE_TM", "32,64,128,256").split(",")]
CFGS = [int(s) for s in os.environ.get("PBN_PROBE_CFG", "0").split(",")]
DT = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[os.environ.get("PBN_PROBE_DTYPE", "bf16")]


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    if STREAMS > 1:
        streams = [torch.cuda.Stream() for _ in range(STREAMS)]
        graphs = []
        for st in streams:
            with torch.cuda.stream(st):
                fn()
            torch.cuda.synchronize()
            gq = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gq, stream=st):
                for _ in range(REP):
                    fn()
            graphs.append(gq)
        def go():
            for st, gq in zip(streams, graphs):
                with torch.cuda.stream(st):
                    gq.replay()
        go(); torch.cuda.synchronize()
        t0 = time.perf_counter(); go(); go(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (2 * REP * STREAMS) * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * REP) * 1e3


tables = {}


def run(level, cin, cout, k=3):
    # k = 3 / 5: cube maps at `level`; k = 2: the k2s2 down convolution from `level` to level + 1; k = -2: the transposed one
    # from level + 1 up to `level`
    if k == 2:
        nbr, n = pyr.down_map(1 << level), pyr.n[level + 1]
        n_in = pyr.n[level]
    elif k == -2:
        nbr, n = pyr.up_map(2 << level), pyr.n[level]
        n_in = pyr.n[level + 1]
    else:
        n = n_in = pyr.n[level]
        nbr = pyr.kernel_map(1 << level, k)
    nbr = nbr.contiguous()
    for tm in TMS:
        if (level, k, tm) in tables:
            continue
        torch.cuda.synchronize()
        HaloTable(nbr, tile_rows=tm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ht = HaloTable(nbr, tile_rows=tm)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / 5 * 1e6
        cnt = ht.counts()
        print("  tables L%d k=%d tile %d: %d rows, halo %.2fx, largest %d, build %.1f us (incl. allocation)" % (
            level, k, tm, n, cnt.sum().item() / n, cnt.max().item(), tb), flush=True) if os.environ.get('PBN_PROBE_VERBOSE') else None
        tables[(level, k, tm)] = ht
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=abs(k), dimension=3).to(dev)
    packed = conv._cache.get(conv.kernel, DT)
    e = 16 // torch.empty(0, dtype=DT).element_size()
    x = torch.randn(n_in, packed[1] * e, device=dev).to(DT)
    out = torch.empty(n, packed[3], dtype=DT, device=dev)
    out2 = torch.empty_like(out)
    t_old = timed(lambda: spconv_forward(x, nbr, n, packed, out=out))
    line = "L%d rows=%6d %3d->%3d K=%3d: gather kernels %.1f us |" % (level, n, cin, cout, abs(k) ** 3, t_old)
    for rpw in [int(v) for v in os.environ.get("PBN_PROBE_OLD_CFG", "").split(",") if v]:
        try:
            line += " old[%d] %.1f" % (rpw, timed(lambda: spconv_forward(x, nbr, n, packed, out=out, rows_per_wave=rpw)))
        except RuntimeError:
            line += " old[%d] n/a" % rpw
    best = None
    for tm in TMS:
      ht = tables[(level, k, tm)]
      for cfg in CFGS:
        for s in SLOTS:
            try:
                t_new = timed(lambda: spconv_forward_halo(x, ht, packed, out=out2, lds_slots=s, cfg=cfg))
            except RuntimeError as ex:
                line += " [tile %d cfg %d slots %d] unsupported" % (tm, cfg, s)
                continue
            err = (out.float() - out2.float()).abs().max().item()
            line += " [%d/%d/%d] %.1f (%.0e)" % (tm, cfg, s, t_new, err)
            if best is None or t_new < best[0]:
                best = (t_new, tm, cfg)
    if best:
        line += "  BEST tile %d cfg %d: %.1f us = x%.2f" % (best[1], best[2], best[0], t_old / best[0])
    print(line, flush=True)


cases = [(0, 96, 96), (0, 128, 96), (1, 96, 96), (1, 128, 96), (1, 32, 32), (2, 64, 64), (2, 128, 128), (2, 192, 128),
         (3, 128, 128), (3, 256, 256), (0, 32, 32, 5)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for c in cases:
    run(*c)
