"""Row-stationary family (csrc/spconv_rs.hip) against the workgroup-tile / wave families on layers of the bench scene:
bit-equality of the outputs (same summation order) and microseconds per launch from a HIP graph (alone, or the same layer on
PBN_PROBE_STREAMS streams at once).  Cases: level,cin,cout[,k] with k = 3 (cube map), 2 (strided down), -2 (transposed up)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward
dev = "cuda:0"
REP = 20
STREAMS = int(os.environ.get("PBN_PROBE_STREAMS", "1"))
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("PBN_PROBE_DTYPE", "bf16")]
CFGS = [int(c) for c in os.environ.get("PBN_PROBE_CFGS", "32,10000").split(",")]
seed = int(os.environ.get("PBN_PROBE_SEED", "2"))
which = os.environ.get("PBN_PROBE_SCENE", "backbone")

batch, _, _ = synth.make_val_batch(seed=seed, copies=1)
coords = torch.from_numpy(batch["xyz_voxel"]).to(dev)
if which != "backbone":                       # a local-scene sized lineage: a 40 % subset of the rows
    keep = torch.rand(coords.shape[0], device=dev) < float(which)
    coords = coords[keep].contiguous()
cm = ME.CoordinateManager(coords)
pyr = cm.sorted().pyramid


def timed(fn_list):
    sts = [torch.cuda.Stream() for _ in fn_list]
    graphs = []
    for st, fn in zip(sts, fn_list):
        with torch.cuda.stream(st):
            for _ in range(2):
                fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(REP):
                fn()
        graphs.append(g)

    def go():
        for st, g in zip(sts, graphs):
            with torch.cuda.stream(st):
                g.replay()
    go(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); go(); go(); torch.cuda.synchronize(); t1 = time.perf_counter()
        best = min(best, (t1 - t0) / (2 * REP * len(fn_list)) * 1e6)
    return best


def run(level, cin, cout, k=3):
    if k == 3:
        n_out = pyr.n[level]; nbr = pyr.kernel_map(1 << level, 3); n_in = n_out; ks = 3
    elif k == 2:
        n_out = pyr.n[level + 1]; nbr = pyr.down_map(1 << level); n_in = pyr.n[level]; ks = 2
    else:
        n_out = pyr.n[level]; nbr = pyr.up_map(1 << (level + 1)); n_in = pyr.n[level + 1]; ks = 2
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=2 if ks == 2 else 1, dimension=3).to(dev)
    x = (torch.randn(n_in, cin, device=dev) * 0.5).to(DT)
    packed = conv._cache.get(conv.kernel, DT)
    cp = packed[3]
    scale = torch.rand(cp, device=dev) + 0.5
    shift = torch.randn(cp, device=dev)
    res = torch.randn(n_out, cp, device=dev).to(DT)
    outs, line = {}, "L%d rows=%6d %3d->%3d K=%2d %s:" % (level, n_out, cin, cout, nbr.shape[1], str(DT).split(".")[-1])
    for cfg in CFGS:
        def call(o, cfg=cfg):
            return spconv_forward(x, nbr, n_out, packed, scale=scale, shift=shift, residual=res, relu=True, out=o, rows_per_wave=cfg)
        try:
            o = torch.zeros(n_out, cp, dtype=DT, device=dev)
            call(o)
            torch.cuda.synchronize()
        except RuntimeError as e:
            line += "  %d: unsupported" % cfg
            continue
        outs[cfg] = o
        bufs = [torch.empty_like(o) for _ in range(STREAMS)]
        t = timed([(lambda b=b: call(b)) for b in bufs])
        line += "  %d: %.1f us" % (cfg, t)
    ref = outs.get(CFGS[0])
    for cfg, o in outs.items():
        if cfg != CFGS[0] and ref is not None:
            same = torch.equal(o, ref)
            line += "  [%d == %d: %s%s]" % (cfg, CFGS[0], same, "" if same else " maxdiff %.3g" % (o.float() - ref.float()).abs().max().item())
    print(line, flush=True)


CASES = os.environ.get("PBN_PROBE_CASES", "0,96,96;0,128,96;1,96,96;1,128,96;1,32,32;0,96,96,-2;0,32,32,2").split(";")
for c in CASES:
    run(*[int(v) for v in c.split(",")])
