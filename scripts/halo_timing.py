"""Cycle stamps of k_spconv_halo on one layer of the bench scene (debug build: make -C pbnet_amd/csrc timing).
usage: halo_timing.py <level> <cin> <cout> <tile_rows> <cfg> [k]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PBNET_HIP_LIB"] = os.path.join(ROOT, "pbnet_amd", "libpbnet_hip_timing.so")
import numpy as np
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import spconv_forward_halo, HaloTable

dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid
lib = ctypes.CDLL(os.environ["PBNET_HIP_LIB"])
HB, HSN = 256, 48


def run(level, cin, cout, tm, cfg, k=3):
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, k)
    ht = HaloTable(nbr, tile_rows=tm)
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3).to(dev)
    packed = conv._cache.get(conv.kernel, torch.bfloat16)
    x = torch.randn(n, packed[1] * 8, device=dev).to(torch.bfloat16)
    out = torch.empty(n, packed[3], dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        spconv_forward_halo(x, ht, packed, out=out, cfg=cfg)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * (HB * 8 * HSN + 8))()
    assert lib.pbn_halo_timing_read(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)[:HB * 8 * HSN].reshape(HB, 8, HSN)
    nw = tm // 32
    nb = min((n + tm - 1) // tm, HB)
    t = a[:nb, :nw]
    d = lambda i, j: (t[:, :, j] - t[:, :, i]).mean()
    print("L%d rows=%d %d->%d K=%d tile %d cfg %d: total %.0f cycles | prologue %.0f | unit list %.0f | first weights issued %.0f | "
          "stage issue %.0f | stage landed %.0f | pipeline prologue %.0f | pass-0 loop %.0f | rest of passes %.0f | epilogue %.0f" % (
              level, n, cin, cout, k ** 3, tm, cfg, d(0, 45), d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(6, 7), d(7, 44), d(44, 45)))
    for j in range(0, 9):
        b = 8 + 4 * j
        if (t[:, :, b + 3] > 0).all():
            print("   iteration %d: wait+barrier %.0f | weight issue %.0f | operands + MFMAs %.0f | (top to top %.0f)" % (
                j, d(b, b + 1), d(b + 1, b + 2), d(b + 2, b + 3), d(b, b + 4) if j < 8 and (t[:, :, b + 4] > 0).all() else float("nan")))


args = [int(v) for v in sys.argv[1:]]
run(*args)
