"""pbn_spconv_wgrad on single layers of the bench scene: exact pair lists (read-back) against the capacity-sized device
lists, microseconds per call (HIP events around 20 calls)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth, _native as N
from pbnet_amd.MinkowskiEngine import conv as C
dev = "cuda:0"
batch, _, _ = synth.make_val_batch(seed=2, copies=1)
cm = ME.CoordinateManager(torch.from_numpy(batch["xyz_voxel"]).to(dev))
pyr = cm.sorted().pyramid


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for level, cin, cout in ((4, 256, 256), (3, 256, 256), (3, 128, 128), (2, 128, 128), (1, 96, 96), (0, 96, 96), (1, 32, 32)):
    n = pyr.n[level]
    nbr = pyr.kernel_map(1 << level, 3)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    g = torch.randn(n, cout, device=dev).to(torch.bfloat16)
    in_idx, out_idx, _, n_seg = C.rulebook_pairs(nbr)
    seg_begin = nbr._pbn_pairs[5]
    lib = N.lib()
    dw = torch.empty(27, cin, cout, dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib.pbn_spconv_wgrad_workspace_bytes(27, cin, cout)), dtype=torch.uint8, device=dev)

    def exact(n_pairs=n_seg * C.WGRAD_PAIR_SEGMENT):
        N.check(lib.pbn_spconv_wgrad(N.c_vp(x.data_ptr()), cin, N.c_vp(g.data_ptr()), cout, 1, N.ptr(in_idx), N.ptr(out_idx),
                                     N.ptr(seg_begin), None, C.WGRAD_PAIR_SEGMENT, n_pairs, 27, cin, cout, N.ptr(dw),
                                     N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream()), "wgrad")
    t_exact = timed(exact)
    ref = dw.clone()
    t_native = timed(lambda: C.wgrad_native(x, g, nbr, cin, cout))
    got = C.wgrad_native(x, g, nbr, cin, cout)
    pairs = int((nbr >= 0).sum())
    print("L%d rows=%6d %3d->%3d pairs %8d (%.2f of the table): exact lists %.1f us, device lists %.1f us, same result %s"
          % (level, n, cin, cout, pairs, pairs / (n * 27), t_exact, t_native, bool(torch.equal(got, ref) or (got - ref).abs().max() < 1e-3 * ref.abs().max())), flush=True)
