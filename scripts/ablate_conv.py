"""Ablation table of the wide-level convolution: one layer of the bench scene under the
PBN_CONV_DBG switches of the kernel (bit 4: no row gathers -- every gather offset out of range; bit 8: no weight DMA; bit 2:
no MFMAs; 16: no main loop at all = prologue + epilogue), alone (HIP-graph replay, HIP events) and with four copies of the layer
on four streams.  The switches are read once per process, so every setting runs in a child process.

    python scripts/ablate_conv.py [out.json]          # driver: spawns the children, writes the table
    python scripts/ablate_conv.py --child LEVEL CIN COUT STREAMS    # one setting (PBN_CONV_DBG / PBN_CONV_FAMILY from the env)
"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

REP = 20


def child(level, cin, cout, streams):
    import torch
    import pbnet_amd.MinkowskiEngine as ME
    from pbnet_amd import synth
    from pbnet_amd.MinkowskiEngine.conv import spconv_forward
    dev = "cuda:0"
    batch, _, _ = synth.make_val_batch(seed=2, copies=1)
    cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
    pyr = cm.sorted().pyramid
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, 3)
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3).to(dev)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    packed = conv._cache.get(conv.kernel, torch.bfloat16)
    rw = int(os.environ.get("PBN_PROBE_RW", "0"))
    sts = [torch.cuda.Stream() for _ in range(streams)]
    outs = [torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev) for _ in range(streams)]
    graphs = []
    for st, o in zip(sts, outs):
        with torch.cuda.stream(st):
            for _ in range(2):
                spconv_forward(x, nbr, n, packed, rows_per_wave=rw, out=o)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(REP):
                spconv_forward(x, nbr, n, packed, rows_per_wave=rw, out=o)
        graphs.append(g)

    def go():
        for st, g in zip(sts, graphs):
            with torch.cuda.stream(st):
                g.replay()
    go(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); go(); go(); torch.cuda.synchronize(); t1 = time.perf_counter()
        best = min(best, (t1 - t0) / (2 * REP * streams) * 1e6)
    print(json.dumps({"rows": int(n), "us_per_launch": round(best, 2)}), flush=True)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_ablate.json"
    layers = {"op52_L0_96_96": (0, 96, 96), "op47_L1_96_96": (1, 96, 96)}
    settings = [("full", 0), ("no_gather", 4), ("no_weight_dma", 8), ("no_mfma", 2), ("no_gather_no_dma", 12),
                ("no_gather_no_mfma", 6), ("no_dma_no_mfma", 10), ("none_of_the_three", 14), ("no_main_loop", 16)]
    fam = os.environ.get("PBN_ABLATE_FAMILIES", "tile").split(",")
    table = {}
    for name, (lv, ci, co) in layers.items():
        for f in fam:
            for sname, dbg in settings:
                for streams in (1, 4):
                    env = dict(os.environ, PBN_CONV_DBG=str(dbg))
                    if f == "tile":
                        env["PBN_CONV_RS"] = "0"
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(lv), str(ci), str(co), str(streams)],
                                       env=env, capture_output=True, text=True)
                    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    val = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
                    table.setdefault(name, {}).setdefault(f, {}).setdefault(sname, {})["alone" if streams == 1 else "four_streams"] = val
                    print(name, f, sname, streams, val, flush=True)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "w") as fh:
        json.dump({"what": "us per launch of one k=3 layer of the bench scene (seed 2) under PBN_CONV_DBG ablations; "
                           "four_streams = four copies on four streams, wall / (launches)", "table": table}, fh, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    else:
        main()
