"""Cycle stamps of k_spconv_wave on one convolution layer of the bench scene (debug build: `make -C pbnet_amd/csrc timing`).
Per segment of the kernel (mean over all stamped workgroups' waves, shader cycles):
  entry -> rulebook tile in LDS | unit list | first loads issued | main loop | partial tiles in LDS | epilogue done
plus the spread of the workgroups' start times (s_memrealtime, 10 ns ticks) over the grid.
usage: wave_timing.py <level> <cin> <cout> <cfg> [k]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PBNET_HIP_LIB"] = os.path.join(ROOT, "pbnet_amd", "libpbnet_hip_timing.so")
import numpy as np
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward

dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
lib = ctypes.CDLL(os.environ["PBNET_HIP_LIB"])
WT = 1024


def run(level, cin, cout, cfg, k=3):
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, k) if k > 1 else None
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3).to(dev)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    packed = conv._cache.get(conv.kernel, torch.bfloat16)
    out = torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev)
    # warm: the same launch a few times (weights and rows L2 / MALL resident), then a cold one behind a large memset
    for mode in ("warm", "cold"):
        for _ in range(3):
            spconv_forward(x, nbr, n, packed, rows_per_wave=cfg, out=out)
        if mode == "cold":
            junk = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
            junk.fill_(1)
            spconv_forward(x, nbr, n, packed, rows_per_wave=cfg, out=out)
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * (WT * 64 + 8))()
        assert lib.pbn_wave_timing_read(buf) == 0
        a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
        gx, gy = int(a[WT * 64]), int(a[WT * 64 + 1])
        nb = min(gx * gy, WT)
        t = a[:WT * 64].reshape(WT, 8, 8)[:nb]
        kw = 8 if cfg >= 1000 else 4
        t = t[:, :kw]
        seg = np.diff(t[:, :, :7], axis=2)
        names = ["tile->LDS", "unit list", "first issue", "main loop", "partials", "epilogue"]
        tot = t[:, :, 6] - t[:, :, 0]
        start = t[:, 0, 7]
        print("L%d rows=%d %d->%d K=%d cfg=%d %s: grid %dx%d | total %.0f cyc (%.2f us @2.4GHz; min %d max %d) | " %
              (level, n, cin, cout, k ** 3, cfg, mode, gx, gy, tot.mean(), tot.mean() / 2400, tot.min(), tot.max()) +
              "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, seg.reshape(-1, 6).mean(0))) +
              " | WG start spread %.2f us (p50 %.2f)" % ((start.max() - start.min()) / 100.0, (np.median(start) - start.min()) / 100.0),
              flush=True)


if len(sys.argv) > 4:
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 3)
else:
    for case in [(4, 256, 256, 1202), (4, 128, 256, 1202, 1), (3, 128, 128, 1204), (3, 256, 256, 1404), (2, 64, 64, 1404),
                 (2, 128, 128, 1404)]:
        run(*case)
