"""Phase counters of the pair-compacted kernel (k_spconv_pc) on one layer of the bench scene (debug build:
`make -C pbnet_amd/csrc experiments`): mean cycles per wave of every phase, over the waves of the first 256 workgroups.
usage: pc_timing.py level cin cout [tile rows = 0 (automatic)] [k = 3 | 2 (down) | -2 (up)]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PBNET_HIP_LIB"] = os.path.join(ROOT, "pbnet_amd", "libpbnet_hip_exp_timing.so")
import numpy as np
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward

dev = "cuda:0"
level, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
cfg = 13000 + 100000 * (rows // 16)
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
if k == 3:
    n = pyr.n[level]; nbr = pyr.kernel_map(1 << level, 3); n_in = n; ks = 3
elif k == 2:
    n = pyr.n[level + 1]; nbr = pyr.down_map(1 << level); n_in = pyr.n[level]; ks = 2
else:
    n = pyr.n[level]; nbr = pyr.up_map(1 << (level + 1)); n_in = pyr.n[level + 1]; ks = 2
torch.manual_seed(0)
conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=2 if ks == 2 else 1, dimension=3).to(dev)
x = torch.randn(n_in, cin, device=dev).to(torch.bfloat16)
packed = conv._cache.get(conv.kernel, torch.bfloat16)
out = torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev)
for _ in range(3):
    spconv_forward(x, nbr, n, packed, rows_per_wave=cfg, out=out)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["PBNET_HIP_LIB"])
buf = (ctypes.c_uint64 * (256 * 8 * 16))()
assert lib.pbn_pc_timing_read(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(256, 8, 16)
names = ["stage + compact", "list, zero tile, meta", "wait own LDS writes", "barrier", "-", "acc reads + MFMA + write-back", "drain + barrier", "epilogue", "weight store / load, gathers, entry reads", "-"]
act = a[a[:, :, :10].sum(2) > 0]
tot = act[:, :10].sum(1).mean()
print("L%d rows=%d %d->%d K=%d tile rows %d: %d waves timed, %.0f cycles per wave" % (level, n, cin, cout, nbr.shape[1], rows, act.shape[0], tot))
for i, nm in enumerate(names):
    v = act[:, i]
    print("  %-24s mean %8.0f  (%.1f %%)  min %8.0f  max %8.0f" % (nm, v.mean(), 100 * v.mean() / tot, v.min(), v.max()))
