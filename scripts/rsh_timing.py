"""Phase counters of the staged row-stationary kernel (k_spconv_rsh) on one layer of the bench scene (debug build:
`make -C pbnet_amd/csrc timing`): mean cycles per wave of every phase, over the waves of the first 256 workgroups."""
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
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 11000
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
buf = (ctypes.c_uint64 * (256 * 8 * 16))()
assert lib.pbn_rsh_timing_read(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(256, 8, 16)
tiles = min(256, (n + 1) // 1)
names = ["rulebook read", "masks", "hash + barrier", "slot table + list", "between passes", "pass barrier", "stage + W0 issue", "batch tail",
         "wait own DMA", "batch barrier", "next W issue", "batch head", "batch steps", "loop exit", "epilogue", "-"]
act = a[a[:, :, :15].sum(2) > 0]
tot = act[:, :15].sum(1).mean()
print("L%d rows=%d %d->%d cfg=%d: %d waves timed, %.0f cycles per wave" % (level, n, cin, cout, cfg, act.shape[0], tot))
for i, nm in enumerate(names[:15]):
    v = act[:, i]
    print("  %-20s mean %8.0f  (%.1f %%)  min %8.0f  max %8.0f" % (nm, v.mean(), 100 * v.mean() / tot, v.min(), v.max()))
