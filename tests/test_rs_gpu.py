"""GPU tier: the row-stationary convolution family (csrc/spconv_rs.hip, round 5) through the C ABI against the CPU oracle
(oracle/sparse_ref.py) -- the same reference arithmetic as pbn_spconv_forward (MinkowskiConvolution forward,
/root/reference/network/Mink.py:221-288,293-350).  Tolerance: 1e-4 ABSOLUTE on fp32 features (BASELINE.json north_star); 16-bit
slabs against the fp32 result of the same rounded inputs with a dtype-sized tolerance.

Covered: the family (bit-identical to the workgroup-tile kernel: same summation order) at every fragment count and several tile
heights, the fused epilogue, ragged last tiles, strided / transposed maps, the folded shortcut (second source), a device-side row
count incl. a count of 0.  (Round 5's staged form and its per-map tables left the library in round 6: spconv_rs.hip.)"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import sparse_ref as R
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coords(seed=47, room=(1.0, 0.8, 0.6)):
    sc = synth.synth_room(seed=seed, pitch=0.0225, room=room, n_boxes=1)
    q, _, _ = synth.voxelize_numpy(sc["xyz"], 0.02)
    return np.concatenate([np.zeros((len(q), 1), np.int32), q], 1).astype(np.int32)


def _close(got, want, what, tol):
    err = (got.float() - want.float()).abs().max().item()
    print("%s: max |diff| %.3e (|want| max %.2f, tol %.1e)" % (what, err, want.abs().max().item(), tol))
    assert err <= tol, "%s: max |diff| %.3e > %.1e" % (what, err, tol)


def _cfg(nf, tile_rows=0):
    """rows_per_wave code of an explicit row-stationary configuration: nf fragments per wave, tile height."""
    return 10000 + 2000 + nf + 100000 * (tile_rows // 16)


def _setup(cin, cout, k, dtype, coords):
    from pbnet_amd.MinkowskiEngine.conv import _pad_vec
    n = len(coords)
    torch.manual_seed(cin * 100 + cout + k)
    feats = torch.randn(n, cin)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3)
    cm_ref = R.CoordinateManager(coords)
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    res = torch.randn(n, cout)
    q = (lambda t: t.to(dtype).float())
    want = R.conv(q(feats), q(conv.kernel.detach()), cm_ref.get_map(1, 1, k), n)
    want = torch.relu(want * scale + shift + q(res))
    conv = conv.to(DEV)
    x = ME.SparseTensor(feats.to(dtype), torch.from_numpy(coords), device=DEV)
    packed = conv._cache.get(conv.kernel, dtype)
    cout_p = packed[3]
    nbr = x.coordinate_manager.kernel_map(1, k)
    sc, sh = _pad_vec(scale.to(DEV), cout_p, 1.0), _pad_vec(shift.to(DEV), cout_p, 0.0)
    resd = torch.zeros(n, cout_p, dtype=dtype, device=DEV)
    resd[:, :cout] = res.to(dtype).to(DEV)
    return x, nbr, n, packed, sc, sh, resd, want


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL), (torch.bfloat16, 6e-2), (torch.float16, 1e-2)])
@pytest.mark.parametrize("cin,cout", [(96, 96), (128, 96), (32, 32), (64, 64), (128, 128), (192, 128), (32, 64)])
def test_row_stationary_forms_against_the_oracle(dtype, tol, cin, cout):
    """Every fragment count, tile heights from one fragment per wave to the largest tile (and heights that leave waves with unequal
    fragment counts and a ragged last tile), with the full fused epilogue; run-to-run bit-identical; against the workgroup-tile
    kernel (scripts/probe_rs.py checks bit-equality on the bench scene's levels)."""
    from pbnet_amd.MinkowskiEngine.conv import spconv_forward
    coords = _coords()
    x, nbr, n, packed, sc, sh, resd, want = _setup(cin, cout, 3, dtype, coords)
    lim = tol if dtype == torch.float32 else tol * max(1.0, want.abs().max().item())
    ref = spconv_forward(x.F, nbr, n, packed, scale=sc, shift=sh, residual=resd, relu=True, rows_per_wave=32)
    ran = 0
    for nf, rows in ((1, 0), (1, 48), (2, 256), (3, 304), (4, 512), (5, 576), (5, 640), (3, 0), (0, 0)):
        cfg = _cfg(nf, rows)
        try:
            o1 = spconv_forward(x.F, nbr, n, packed, scale=sc, shift=sh, residual=resd, relu=True, rows_per_wave=cfg)
        except RuntimeError as e:
            assert "UNSUPPORTED" in str(e), e          # (e.g. the 128-channel shape at 5 fragments is not built for fp32)
            continue
        o2 = spconv_forward(x.F, nbr, n, packed, scale=sc, shift=sh, residual=resd, relu=True, rows_per_wave=cfg)
        assert torch.equal(o1, o2), "cfg %d not deterministic" % cfg
        # same summation order as k_spconv (bit-identical when that launch is not split over K, as on the wide levels this family
        # serves; on this small scene k_spconv splits: fp32 re-association only)
        _close(o1.float().cpu(), ref.float().cpu(), "row-stationary vs k_spconv cfg %d" % cfg, 1e-5 if dtype == torch.float32 else lim)
        _close(o1[:, :cout].float().cpu(), want, "rs nf %d rows %d %d->%d %s" % (nf, rows, cin, cout, dtype), lim)
        ran += 1
    # (the 64- and 128-channel shapes of round 6 are built for 16-bit slabs only, at up to 3 fragments per wave for 128 channels)
    assert ran >= (0 if (dtype == torch.float32 and cout in (64, 128)) else (4 if cout == 128 else 6))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("kind", ["down", "up"])
def test_strided_and_transposed_maps(dtype, tol, kind):
    """k = 2, s = 2 maps (8 offsets, input level != output level)."""
    from pbnet_amd.MinkowskiEngine.conv import spconv_forward
    coords = _coords(seed=50, room=(1.2, 1.0, 0.6))
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(DEV))
    cm_ref = R.CoordinateManager(coords)
    n0, n1 = cm.num_rows(1), cm.num_rows(2)
    cin, cout = 96, 96
    torch.manual_seed(5)
    if kind == "down":
        nbr, n_in, n_out = cm.down_map(1), n0, n1
        conv = ME.MinkowskiConvolution(cin, cout, kernel_size=2, stride=2, dimension=3)
        maps = cm_ref.get_map(1, 2, 2)
    else:
        nbr, n_in, n_out = cm.up_map(2), n1, n0
        conv = ME.MinkowskiConvolutionTranspose(cin, cout, kernel_size=2, stride=2, dimension=3)
        maps = [(o, i) for (i, o) in cm_ref.get_map(1, 2, 2)]
    feats = torch.randn(n_in, cin)
    q = (lambda t: t.to(dtype).float())
    want = R.conv(q(feats), q(conv.kernel.detach()), maps, n_out)
    conv = conv.to(DEV)
    packed = conv._cache.get(conv.kernel, dtype)
    xd = feats.to(dtype).to(DEV)
    lim = tol if dtype == torch.float32 else tol * max(1.0, want.abs().max().item())
    for cfg in (_cfg(0), _cfg(3, 304), _cfg(2, 256)):
        o = spconv_forward(xd, nbr, n_out, packed, rows_per_wave=cfg)
        _close(o[:, :cout].float().cpu(), want, "%s map cfg %d %s" % (kind, cfg, dtype), lim)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL), (torch.bfloat16, 6e-2), (torch.float16, 1e-2)])
def test_folded_shortcut_on_the_row_stationary_family(dtype, tol):
    """pbn_spconv_forward_dual (a BasicBlock's 1x1 shortcut as reduction steps of its second convolution, Mink.py:77-87) against the
    oracle's two convolutions."""
    from pbnet_amd.MinkowskiEngine.conv import spconv_forward_dual, pack_weight, _pad_vec
    from pbnet_amd.network.mink_unet import _group_steps
    coords = _coords(seed=49)
    n = len(coords)
    cin, cin2, cout = 96, 128, 96
    torch.manual_seed(11)
    h, x = torch.randn(n, cin), torch.randn(n, cin2)
    k2, kd = torch.randn(27, cin, cout) * (2.0 / (27 * cin)) ** 0.5, torch.randn(1, cin2, cout) * (1.0 / cin2) ** 0.5
    s2, b2, sd, bd = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1, torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    q = (lambda t: t.to(dtype).float())
    cm_ref = R.CoordinateManager(coords)
    want = torch.relu(R.conv(q(h), q(k2 * s2), cm_ref.get_map(1, 1, 3), n) + b2 + q(x) @ q(kd[0] * sd) + bd)
    st = ME.SparseTensor(torch.zeros(n, 1), torch.from_numpy(coords), device=DEV)
    nbr = st.coordinate_manager.kernel_map(1, 3)
    w2, vpo, n_main, cout_p = pack_weight((k2 * s2).to(DEV), dtype)
    wd, vpo2, n2, _ = pack_weight((kd * sd).to(DEV), dtype)
    pad = (-n2) % _group_steps(vpo // 4)
    w = torch.cat([w2, wd] + ([torch.zeros(pad, *wd.shape[1:], dtype=wd.dtype, device=DEV)] if pad else []), 0).contiguous()
    e = 16 // torch.empty(0, dtype=dtype).element_size()
    hd = torch.zeros(n, vpo * e, dtype=dtype, device=DEV); hd[:, :cin] = h.to(dtype).to(DEV)
    xd = torch.zeros(n, vpo2 * e, dtype=dtype, device=DEV); xd[:, :cin2] = x.to(dtype).to(DEV)
    shift = _pad_vec((b2 + bd).to(DEV), cout_p, 0.0)
    lim = tol if dtype == torch.float32 else tol * max(1.0, want.abs().max().item())
    ran = 0
    for cfg in (_cfg(0), _cfg(4, 512), _cfg(3, 304), _cfg(2, 160)):
        try:
            o1 = spconv_forward_dual(hd, nbr, n, xd, (w, vpo, n_main + n2 + pad, cout_p), vpo2, shift=shift, relu=True, rows_per_wave=cfg)
        except RuntimeError as ex:
            assert "UNSUPPORTED" in str(ex), ex
            continue
        o2 = spconv_forward_dual(hd, nbr, n, xd, (w, vpo, n_main + n2 + pad, cout_p), vpo2, shift=shift, relu=True, rows_per_wave=cfg)
        assert torch.equal(o1, o2)
        _close(o1[:, :cout].float().cpu(), want, "dual cfg %d %s" % (cfg, dtype), lim)
        ran += 1
    assert ran >= 3


def test_device_side_row_count_bounds_the_launch():
    """n_out as a capacity + the row count on the device (the planned forward's form): rows past the count are not written."""
    import ctypes
    from pbnet_amd import _native as N
    from pbnet_amd.MinkowskiEngine.conv import _DT, _workspace, spconv_forward
    coords = _coords(seed=51)
    dtype = torch.bfloat16
    x, nbr, n, packed, sc, sh, resd, want = _setup(96, 96, 3, dtype, coords)
    w, vpo, n_steps, cout_p = packed
    cap = n + 300
    nbr_cap = torch.full((cap, 27), 123456, dtype=torch.int32, device=DEV)        # garbage behind the real rows
    nbr_cap[:n] = nbr
    feats = torch.zeros(cap, x.F.shape[1], dtype=dtype, device=DEV); feats[:n] = x.F
    n_dev = torch.tensor([n], dtype=torch.int32, device=DEV)
    ws = _workspace(torch.device(DEV))
    for cfg in (_cfg(0), _cfg(2, 176)):
        out = torch.full((cap, cout_p), 7.0, dtype=dtype, device=DEV)
        rc = N.lib().pbn_spconv_forward(
            N.c_vp(feats.data_ptr()), feats.stride(0), cap, N.c_vp(nbr_cap.data_ptr()), 27, None, N.c_vp(n_dev.data_ptr()), cap,
            N.c_vp(w.data_ptr()), vpo, n_steps, cout_p, N.c_vp(sc.data_ptr()), N.c_vp(sh.data_ptr()), None, 0, 1,
            N.c_vp(out.data_ptr()), out.stride(0), _DT[dtype], cfg, N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream())
        N.check(rc, "pbn_spconv_forward")
        assert torch.all(out[n:] == 7.0), "rows past the device-side count were written (cfg %d)" % cfg
        exact = spconv_forward(x.F, nbr, n, packed, scale=sc, shift=sh, relu=True, rows_per_wave=32)
        assert torch.allclose(out[:n].float(), exact.float(), atol=6e-2 * max(1.0, want.abs().max().item()))
        # a device-side count of 0 (an empty mask / score lineage over a capacity): nothing is written, nothing divides by zero
        zero = torch.zeros(1, dtype=torch.int32, device=DEV)
        out = torch.full((cap, cout_p), 7.0, dtype=dtype, device=DEV)
        rc = N.lib().pbn_spconv_forward(
            N.c_vp(feats.data_ptr()), feats.stride(0), cap, N.c_vp(nbr_cap.data_ptr()), 27, None, N.c_vp(zero.data_ptr()), cap,
            N.c_vp(w.data_ptr()), vpo, n_steps, cout_p, N.c_vp(sc.data_ptr()), N.c_vp(sh.data_ptr()), None, 0, 1,
            N.c_vp(out.data_ptr()), out.stride(0), _DT[dtype], cfg, N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream())
        N.check(rc, "pbn_spconv_forward")
        torch.cuda.synchronize()
        assert torch.all(out == 7.0), "a device-side count of 0 wrote rows (cfg %d)" % cfg
