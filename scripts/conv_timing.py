"""Cycle stamps of k_spconv's main loop on one convolution layer of the bench scene (debug build: `make -C pbnet_amd/csrc
timing`).  Prints, for the middle workgroup's four waves, the mean cycles of each segment of a group:
  wait DMA | barrier | DMA issue + next rows | wait chunk 0 | chunks 0..last-1 | last chunk (MFMA + refill) | loop edge."""
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
level, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 32
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
n = pyr.n[level]
nbr = pyr.kernel_map(1 << level, 3)
torch.manual_seed(0)
conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3).to(dev)
x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
packed = conv._cache.get(conv.kernel, torch.bfloat16)
out = torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev)
for _ in range(3):
    spconv_forward(x, nbr, n, packed, rows_per_wave=cfg, out=out)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["PBNET_HIP_LIB"])
buf = (ctypes.c_uint32 * (4 * 64 * 8 + 8))()
assert lib.pbn_conv_timing_read(buf) == 0
a = np.frombuffer(buf, dtype=np.uint32).astype(np.int64)
ng, cg = int(a[2048]), int(a[2049])
t = a[:2048].reshape(4, 64, 8)
m = min(ng, 64)
print("L%d rows=%d %d->%d cfg=%d: %d groups in the timed workgroup, cg=%d" % (level, n, cin, cout, cfg, ng, cg))
names = ["wait DMA", "barrier", "DMA+rows", "wait x0", "chunks", "last chunk", "loop edge"]
for w in range(4):
    tw = t[w, :m]
    seg = np.diff(tw[:, :7], axis=1) & 0xffffffff
    edge = (tw[1:, 0] - tw[:-1, 6]) & 0xffffffff
    per_group = (tw[1:, 0] - tw[:-1, 0]) & 0xffffffff
    print("wave %d: group %.0f cyc (min %d max %d) | " % (w, per_group[2:].mean(), per_group[2:].min(), per_group[2:].max()) +
          "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, list(seg[2:].mean(0)) + [edge[2:].mean()])))
print("total main loop (wave 0): %d cycles for %d groups" % ((t[0, m - 1, 6] - t[0, 0, 0]) & 0xffffffff, m))
