"""GPU tier: the LDS-staged ("halo") convolution family (csrc/spconv_halo.hip) -- the tables of pbn_halo_build against a
numpy statement of the same thing, and the convolution over them against the oracle (oracle/sparse_ref.py) with the full
fused epilogue, on Z-ordered and on caller-ordered maps, with the LDS row buffer cut small (several segments per tile) and
with tiles marked for the plain gather loop.  fp32: 1e-4 absolute (BASELINE.json north_star); bf16 / f16: dtype-sized."""
import numpy as np
import pytest
import torch

from oracle import sparse_ref as R
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.MinkowskiEngine.conv import HaloTable, spconv_forward, spconv_forward_halo, _pad_vec

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _coords(seed, room=(1.0, 0.8, 0.6), n_boxes=1):
    sc = synth.synth_room(seed=seed, pitch=0.0225, room=room, n_boxes=n_boxes)
    q, _, _ = synth.voxelize_numpy(sc["xyz"], 0.02)
    return np.concatenate([np.zeros((len(q), 1), np.int32), q], 1).astype(np.int32)


def _maps(coords, k, order):
    """(nbr on the device, the pyramid's rows as a numpy permutation of the input rows)"""
    x = ME.SparseTensor(torch.zeros(len(coords), 1), torch.from_numpy(coords), device=DEV)
    cm = x.coordinate_manager
    if order == "sorted":
        sv = cm.sorted()
        return sv.pyramid.kernel_map(1, k).contiguous(), sv.perm.cpu().numpy()
    return cm.kernel_map(1, k).contiguous(), np.arange(len(coords))


@pytest.mark.parametrize("k,tm", [(3, 128), (5, 128), (3, 256), (3, 32), (3, 64)])
@pytest.mark.parametrize("order", ["sorted", "plain"])
def test_halo_tables_match_numpy(k, tm, order):
    coords = _coords(51)
    nbr, _ = _maps(coords, k, order)
    n, K = nbr.shape
    ht = HaloTable(nbr, tile_rows=tm)
    L = ht.layout
    assert L.tile_rows == tm and L.n_offsets == K and L.tiles == (n + tm - 1) // tm
    cnt = ht.counts().cpu().numpy()
    rows = ht.view("rows", torch.int32, L.tiles * L.pitch).cpu().numpy().reshape(L.tiles, L.pitch)
    loc = ht.view("loc", torch.int16, L.tiles * tm * K).cpu().numpy().view(np.uint16).reshape(L.tiles, tm, K)
    fm = ht.view("fmask", torch.int16, L.tiles * K).cpu().numpy().view(np.uint16).reshape(L.tiles, K)
    h = nbr.cpu().numpy()
    tot = 0
    for t in range(L.tiles):
        blk = h[t * tm:(t + 1) * tm]
        u = np.unique(blk[blk >= 0])
        assert cnt[t] == len(u), (t, cnt[t], len(u))
        assert np.array_equal(rows[t, :len(u)], u)
        want = np.full((tm, K), 0xffff, np.uint16)
        want[:len(blk)] = np.where(blk >= 0, np.searchsorted(u, np.maximum(blk, 0)), 0xffff).astype(np.uint16)
        assert np.array_equal(loc[t], want)
        wm = np.zeros(K, np.uint16)
        for f in range((len(blk) + 15) // 16):
            wm |= ((blk[f * 16:(f + 1) * 16] >= 0).any(0).astype(np.uint16) << np.uint16(f))
        assert np.array_equal(fm[t], wm)
        tot += len(u)
    print("k=%d tile %d %s: %d rows, %d tiles, halo %.2fx, largest %d" % (k, tm, order, n, L.tiles, tot / n, cnt.max()))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL), (torch.bfloat16, 6e-2), (torch.float16, 1e-2)])
@pytest.mark.parametrize("cin,cout,k,order", [(32, 32, 3, "sorted"), (96, 96, 3, "sorted"), (128, 96, 3, "sorted"),
                                              (64, 128, 3, "plain"), (32, 32, 5, "sorted"), (40, 64, 5, "sorted"),
                                              (192, 256, 3, "sorted")])
def test_halo_convolution_matches_oracle(dtype, tol, cin, cout, k, order):
    coords = _coords(53)
    n = len(coords)
    torch.manual_seed(cin * 100 + cout + k)
    feats = torch.randn(n, cin)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3)
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    res = torch.randn(n, cout)
    q = (lambda t: t.to(dtype).float())
    want = R.conv(q(feats), q(conv.kernel.detach()), R.CoordinateManager(coords).get_map(1, 1, k), n)
    want = torch.relu(want * scale + shift + q(res))
    conv = conv.to(DEV)
    nbr, perm = _maps(coords, k, order)
    packed = conv._cache.get(conv.kernel, dtype)
    w, vpo, n_steps, cout_p = packed
    e = 16 // torch.empty(0, dtype=dtype).element_size()
    x = torch.zeros(n, vpo * e, dtype=dtype, device=DEV)
    x[:, :cin] = feats[perm].to(dtype).to(DEV)
    sc, sh = _pad_vec(scale.to(DEV), cout_p, 1.0), _pad_vec(shift.to(DEV), cout_p, 0.0)
    resd = torch.zeros(n, cout_p, dtype=dtype, device=DEV)
    resd[:, :cout] = res[perm].to(dtype).to(DEV)
    lim = tol if dtype == torch.float32 else tol * max(1.0, want.abs().max().item())
    want_p = want[perm]
    tabs = {tm: HaloTable(nbr, tile_rows=tm) for tm in (32, 64, 128, 256)}
    full, full256 = tabs[128], tabs[256]
    ref = spconv_forward(x, nbr, n, packed, scale=sc, shift=sh, residual=resd, relu=True)      # the round-1..3 kernels
    spo = vpo // 4
    # the wave-autonomous family with staged rows (csrc/spconv_wave_halo.hip): automatic configuration per tile height, small
    # LDS buffers (several segments per tile), marked tiles (plain gather loop), every built configuration
    cases = [("tile %d" % tm, tabs[tm], 0, 0) for tm in (32, 64, 128, 256)]
    cases += [("48-slot buffer", full, 48, 0), ("16-slot buffer, tile 256", full256, 16, 0), ("32-slot buffer, tile 32", tabs[32], 32, 0),
              ("gather loop", HaloTable(nbr, max_rows=100), 0, 0), ("gather loop, tile 64", HaloTable(nbr, tile_rows=64, max_rows=60), 0, 0)]
    ntt = cout_p // 16
    for code in (402, 404, 406, 408, 202, 204, 206, 208, 1401, 1402, 1404, 1408, 1201, 1202, 1204, 1208):
        nf, nt = (code // 100) % 10, code % 100
        if ntt % nt:
            continue
        tm = nf * 16 if code >= 1000 else nf * 64
        for depth in (2, 3):
            cases.append(("cfg %d depth %d" % (code, depth), tabs[tm], 0, 10000 * depth + code))
    # the barrier-synchronised experiment (csrc/spconv_halo.hip): -(100 * units per iteration + 10 * ring slots + steps per pass)
    for s_, r_, c_ in ((3, 2, 1), (4, 2, 1), (4, 2, 2), (3, 2, 3), (4, 2, 4)):
        if spo % c_ == 0:
            cases.append(("barrier cfg %d%d%d" % (s_, r_, c_), full if (s_ + c_) % 2 else full256, 0, -(100 * s_ + 10 * r_ + c_)))
    outs = []
    for what, ht, slots, cfg in cases:
        if what.startswith("gather loop"):
            assert int((ht.counts() < 0).sum().item()) > 0
        try:
            o1 = spconv_forward_halo(x, ht, packed, scale=sc, shift=sh, residual=resd, relu=True, lds_slots=slots, cfg=cfg)
        except RuntimeError as ex:                 # a shape whose buffers exceed the LDS (or a configuration that is not built)
            assert "UNSUPPORTED" in str(ex) and (cfg != 0 or k == 5), (what, ex)
            continue
        o2 = spconv_forward_halo(x, ht, packed, scale=sc, shift=sh, residual=resd, relu=True, lds_slots=slots, cfg=cfg)
        assert torch.equal(o1, o2), what + ": not deterministic"
        err = (o1[:, :cout].float().cpu() - want_p).abs().max().item()
        print("%d->%d k=%d %s %s [%s]: max |diff| %.3e (tol %.1e), vs the gather kernels %.3e" % (
            cin, cout, k, order, dtype, what, err, lim, (o1.float() - ref.float()).abs().max().item()))
        assert err <= lim, (what, err)
        outs.append(o1)


def test_halo_capacity_and_device_count():
    """n_out as a capacity with the live count on the device: rows at and beyond it are neither read nor written."""
    import ctypes
    from pbnet_amd import _native as N
    coords = _coords(55)
    nbr, perm = _maps(coords, 3, "sorted")
    n = nbr.shape[0]
    live = n - 77
    nb = nbr[:live].clone()
    nb[nb >= live] = -1
    cap = torch.full((n, 27), 12345, dtype=torch.int32, device=DEV)     # garbage behind the live rows
    cap[:live] = nb
    n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
    lay = N.HaloLayout()
    nbytes = N.lib().pbn_halo_bytes(n, 27, 128, ctypes.byref(lay))
    table = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    job = N.HaloJob()
    job.nbr = cap.data_ptr(); job.n_out_dev = n_dev.data_ptr(); job.table = table.data_ptr(); job.layout = lay; job.n_out = n
    N.check(N.lib().pbn_halo_build((N.HaloJob * 1)(job), 1, N.current_stream()), "pbn_halo_build")
    conv = ME.MinkowskiConvolution(32, 32, kernel_size=3, dimension=3).to(DEV)
    packed = conv._cache.get(conv.kernel, torch.float32)
    w, vpo, n_steps, cout_p = packed
    x = torch.randn(n, 32, device=DEV)
    out = torch.full((n, cout_p), 7.0, device=DEV)
    rc = N.lib().pbn_spconv_forward_halo(N.c_vp(x.data_ptr()), 32, n, N.c_vp(cap.data_ptr()), 27, N.c_vp(n_dev.data_ptr()), n,
                                         N.c_vp(w.data_ptr()), vpo, n_steps, cout_p, None, None, None, 0, 0,
                                         N.c_vp(out.data_ptr()), cout_p, 0, N.c_vp(table.data_ptr()), ctypes.byref(lay), 0, 0,
                                         N.current_stream())
    N.check(rc, "pbn_spconv_forward_halo")
    want = spconv_forward(x[:live].contiguous(), nb.contiguous(), live, packed)
    assert (out[:live] - want).abs().max().item() <= TOL
    assert bool((out[live:] == 7.0).all())
