"""GPU tier: pbn_spconv_wgrad (csrc/wgrad.hip) on its own -- the register-ring kernel (default) against the round-2 kernel
(PBN_WGRAD_FORM is read once per process, so the cross-check runs in a child process) and against a float64 contraction of
the same pair lists: every tile shape of the MinkUNet layers, channel tails, identity pairs (1x1 / linear), empty offsets,
pair ranges that are not a multiple of the step, strided slab views (a skip slab's columns)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _map(n_out, n_in, k, fill, seed, empty=()):
    g = torch.Generator().manual_seed(seed)
    nbr = torch.randint(0, n_in, (n_out, k), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, k, generator=g) > fill] = -1
    for e in empty:
        nbr[:, e] = -1
    return nbr


def _reference(x, g, nbr, cin, cout):
    k = 1 if nbr is None else nbr.shape[1]
    out = torch.zeros(k, cin, cout, dtype=torch.float64)
    xd, gd = x.double().cpu(), g.double().cpu()
    if nbr is None:
        out[0] = xd[:, :cin].t() @ gd[:, :cout]
        return out
    nb = nbr.cpu().long()
    for o in range(k):
        rows = torch.nonzero(nb[:, o] >= 0).flatten()
        if len(rows):
            out[o] = xd[nb[rows, o]][:, :cin].t() @ gd[rows][:, :cout]
    return out


CASES = [  # n_out, n_in, K, cin, cout, fill, ld_x extra, ld_g extra
    (3000, 3000, 27, 32, 32, 0.5, 0, 0),
    (3000, 3000, 27, 64, 64, 0.5, 0, 0),
    (2500, 2500, 27, 96, 96, 0.4, 0, 0),
    (2500, 2500, 27, 128, 96, 0.4, 0, 0),
    (1500, 1500, 27, 128, 128, 0.6, 0, 0),
    (800, 800, 27, 256, 256, 0.6, 0, 0),
    (700, 5000, 8, 256, 128, 0.9, 0, 0),        # k2s2 down
    (5000, 700, 8, 128, 96, 0.125, 32, 0),      # transposed: one parent per row; x is a view of a wider slab
    (4000, 4000, 125, 6, 32, 0.2, 2, 64),       # stem: 6 channels in an 8-wide slab, output into a skip slab's columns
    (2000, 2000, 27, 48, 80, 0.5, 0, 0),        # odd tile counts
    (1000, 1000, 27, 32, 20, 0.5, 0, 4),        # channel tail on the output side
    (37, 37, 27, 64, 64, 0.5, 0, 0),            # fewer pairs than one step
    (5000, 5000, 27, 64, 32, 0.3, 0, 0),
]


def _run_cases(dtype):
    from pbnet_amd.MinkowskiEngine import conv as C
    res = []
    for ci, (n_out, n_in, k, cin, cout, fill, ex, eg) in enumerate(CASES):
        torch.manual_seed(100 + ci)
        ldx, ldg = (cin + 7) // 8 * 8 + ex, (cout + 7) // 8 * 8 + eg
        xs = torch.zeros(n_in, ldx)
        xs[:, :cin] = torch.randn(n_in, cin)
        gs = torch.zeros(n_out, ldg)
        gs[:, :cout] = torch.randn(n_out, cout)
        xs, gs = xs.to(dtype).to(DEV), gs.to(dtype).to(DEV)
        x = xs[:, :(cin + 7) // 8 * 8] if ex else xs
        g = gs[:, :(cout + 7) // 8 * 8] if eg else gs
        nbr = _map(n_out, n_in, k, fill, 7 + ci, empty=(3,) if k == 27 else ()).to(DEV)
        got = C.wgrad_native(x, g, nbr, cin, cout)
        again = C.wgrad_native(x, g, nbr, cin, cout)
        res.append((got.cpu(), bool(torch.equal(got, again)), _reference(xs, gs, nbr, cin, cout)))
    # identity pairs (1x1 convolution / linear layer)
    for cin, cout, n in ((96, 32, 5000), (256, 256, 700), (32, 3, 9000)):
        torch.manual_seed(cin + n)
        x = torch.randn(n, cin).to(dtype).to(DEV)
        g = torch.zeros(n, (cout + 7) // 8 * 8)
        g[:, :cout] = torch.randn(n, cout)
        g = g.to(dtype).to(DEV)
        got = C.wgrad_native(x, g, None, cin, cout)
        res.append((got.cpu(), bool(torch.equal(got, C.wgrad_native(x, g, None, cin, cout))), _reference(x, g, None, cin, cout)))
    return res


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wgrad_against_float64(dtype):
    for ci, (got, same, ref) in enumerate(_run_cases(dtype)):
        assert same, "case %d: not deterministic" % ci
        err = (got.double() - ref).abs().max().item()
        # products of 16-bit operands are exact in fp32; fp32 accumulation of up to ~1e5 terms of magnitude ~1
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()) * 4, "case %d: max error %.3e" % (ci, err)
        if ci < len(CASES) and CASES[ci][2] == 27:
            assert float(got[3].abs().max()) == 0.0          # an offset without pairs: exact zeros


def test_wgrad_ring_matches_round2_kernel():
    """The same lists through k_wgrad16 (PBN_WGRAD_FORM=16) in a child process: same tiles, same pair order inside a
    workgroup -> the two kernels differ only by where the pair range is split."""
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_wgrad_gpu as T\n"
            "res = T._run_cases(torch.bfloat16)\n"
            "torch.save([r[0] for r in res], sys.argv[1])\n" % (ROOT, os.path.join(ROOT, "tests")))
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "wgrad_form16_%d.pt" % os.getpid())
    env = dict(os.environ, PBN_WGRAD_FORM="16")
    subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=600)
    old = torch.load(path)
    os.remove(path)
    new = _run_cases(torch.bfloat16)
    for ci, (o, (n, _, ref)) in enumerate(zip(old, new)):
        scale = max(1.0, ref.abs().max().item())
        assert (o - n).abs().max().item() <= 1e-4 * scale, "case %d" % ci


def test_wgrad_checked_refuses_what_the_kernels_cannot_address():
    """pbn_spconv_wgrad_checked (round 4; the entry the package calls): slabs of 4 GiB, 2^30 pairs -> PBN_ERR_RANGE; device-built
    lists without their pair counts, more identity pairs than rows -> PBN_ERR_ARG.  Refused before any launch (the pointers are
    never touched) -- and the same call with honest sizes runs."""
    import ctypes
    from pbnet_amd import _native as N
    lib = N.lib()
    x = torch.randn(64, 32, device=DEV).to(torch.bfloat16)
    g = torch.randn(64, 16, device=DEV).to(torch.bfloat16)
    dw = torch.empty(1, 32, 16, dtype=torch.float32, device=DEV)
    ws = torch.empty(int(lib.pbn_spconv_wgrad_workspace_bytes(1, 32, 16)), dtype=torch.uint8, device=DEV)
    idx = torch.arange(64, dtype=torch.int32, device=DEV)
    seg = torch.tensor([0, 1], dtype=torch.int32, device=DEV)

    def call(n_x, n_g, in_idx=None, out_idx=None, seg_begin=None, counts=None, padded=0, n_pairs=64):
        return lib.pbn_spconv_wgrad_checked(N.c_vp(x.data_ptr()), 32, n_x, N.c_vp(g.data_ptr()), 16, n_g, 1, N.ptr(in_idx),
                                            N.ptr(out_idx), N.ptr(seg_begin), N.ptr(counts), padded, 4096 if seg_begin is not None else 0,
                                            n_pairs, 1, 32, 16, N.ptr(dw), N.c_vp(ws.data_ptr()), ws.numel(), N.current_stream())
    assert call((1 << 32) // 64, 64) == N.PBN_ERR_RANGE                 # x slab: rows * 32 * 2 bytes = 4 GiB
    assert call(64, (1 << 32) // 32) == N.PBN_ERR_RANGE                 # g slab
    assert call(64, 64, n_pairs=1 << 30) == N.PBN_ERR_RANGE
    assert call(64, 64, idx, idx, seg, None, 0) == N.PBN_ERR_ARG        # unpadded lists without their counts
    assert call(32, 64, n_pairs=64) == N.PBN_ERR_ARG                    # identity pairs beyond the rows
    assert call(64, 64) == 0
    torch.cuda.synchronize()
    want = x.float().t() @ g.float()
    assert (dw[0] - want).abs().max().item() <= 1e-3
