"""Cycle stamps of k_wgrad_ring's main loop (debug library: make -C pbnet_amd/csrc timing, PBNET_HIP_LIB=pbnet_amd/libpbnet_hip_timing.so):
wave 0 of workgroup 0, first 32 steps; columns = 100 MHz ticks between the stamps (wait, barrier, index DMA, gathers, compute)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth, _native as N
from pbnet_amd.MinkowskiEngine import conv as C
dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
lib = N.lib()
for level, cin, cout, wgs in ((2, 128, 128, 1), (2, 128, 128, 1024), (1, 32, 32, 1), (0, 96, 96, 2048)):
    os.environ["PBN_WGRAD_WGS"] = str(wgs)
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, 3)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    g = torch.randn(n, cout, device=dev).to(torch.bfloat16)
    for _ in range(3):
        C.wgrad_native(x, g, nbr, cin, cout)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (32 * 6))()
    N.check(lib.pbn_wgrad_timing_read(buf), "timing")
    t = [[buf[s * 6 + i] for i in range(6)] for s in range(32)]
    print("L%d %d->%d wgs %d: step: wait to_lds issue reads mfma | step period" % (level, cin, cout, wgs))
    for s in range(4, 20):
        d = [t[s][i + 1] - t[s][i] for i in range(5)]
        print("  %2d: %5d %5d %5d %5d %5d | %6d" % (s, d[0], d[1], d[2], d[3], d[4], t[s + 1][0] - t[s][0]))
