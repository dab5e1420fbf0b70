"""GPU tier: the native training executor of the MinkUNet body (network/train_engine.py over csrc/train_exec.hip) against the
module path it replaces (one autograd node per block, MinkowskiEngine/fused_train.py): same kernels in the same order, so
outputs, running statistics and -- in fp32 -- every gradient agree to the last bit.  With 16-bit slabs the two differ in how
a gradient with two producers is rounded: where an encoder output feeds both the next down convolution and a skip, autograd
adds two ROUNDED gradients (round(round(a) + round(b))) while the executor adds in the convolution's fp32 epilogue
(round(a + round(b))); a block with a 1x1 shortcut adds its two input gradients in the epilogue on both paths, but in the
opposite order.  The last stage sees identical gradients, the layers before it agree to a few roundings."""
import numpy as np
import pytest
import torch

import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.network import train_engine as TE
from pbnet_amd.network.Mink import Mink_unet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _coords(seed=61, batch=2):
    sc = synth.synth_room(seed=seed, pitch=0.0225, room=(0.7, 0.5, 0.4), n_boxes=1)
    q, _, _ = synth.voxelize_numpy(sc["xyz"], 0.02)
    parts = [np.concatenate([np.full((len(q), 1), b, np.int32), q + np.array([9 * b, 0, 0], np.int32)], 1) for b in range(batch)]
    return np.concatenate(parts, 0).astype(np.int32)


def _step(net, feats, coords, target, engine, want_dx, sorted_rows=False):
    TE.ENABLED, TE.SORTED = engine, sorted_rows
    for p in net.parameters():
        p.grad = None
    x = feats.clone().requires_grad_(want_dx)
    out = net(ME.SparseTensor(x, torch.from_numpy(coords).to(DEV))).F
    loss = ((out.float() - target) ** 2).mean()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    bufs = {k: b.detach().clone() for k, b in net.named_buffers()}
    return out.detach().clone(), (x.grad.detach().clone() if want_dx else None), grads, bufs


@pytest.mark.parametrize("arch,dtype,want_dx", [("MinkUNet14A", torch.bfloat16, True), ("MinkUNet34C", torch.bfloat16, False),
                                                ("MinkUNet14A", torch.float32, True), ("MinkUNet18A", torch.float16, False)])
def test_engine_matches_module_path(arch, dtype, want_dx):
    coords = _coords()
    torch.manual_seed(5)
    net = Mink_unet(6, 32, arch=arch).to(DEV).train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    feats = torch.randn(len(coords), 6, device=DEV).to(dtype)
    target = torch.randn(len(coords), 32, device=DEV)
    try:
        ref = _step(net, feats, coords, target, False, want_dx)
        net.load_state_dict(state)
        got = _step(net, feats, coords, target, True, want_dx)
        assert net.__dict__.get("_train_plans"), "the executor did not run"
        net.load_state_dict(state)
        zord = _step(net, feats, coords, target, True, want_dx, sorted_rows=True) if dtype == torch.float32 else None
    finally:
        TE.ENABLED, TE.SORTED = True, False
    assert torch.equal(got[0], ref[0])
    for k in ref[3]:
        assert torch.equal(got[3][k], ref[3][k]), k

    def close(a, b, what):
        a, b = a.float(), b.float()
        # one extra bf16 rounding (2^-9) per skip, carried through up to ~20 layers of 16-bit gradients; fp32 is compared exactly
        assert (a - b).norm().item() <= 4e-2 * max(b.norm().item(), 1e-12), what

    exact_everywhere = dtype == torch.float32
    behind_skips = ("block8", "final_sematic")
    if want_dx:
        assert torch.equal(got[1], ref[1]) if exact_everywhere else close(got[1], ref[1], "input gradient") is None
    n_exact = 0
    for k in ref[2]:
        if exact_everywhere or k.startswith(behind_skips):
            assert torch.equal(got[2][k], ref[2][k]), k
            n_exact += 1
        else:
            close(got[2][k], ref[2][k], k)
    assert n_exact >= 8
    # PBN_TRAIN_SORTED: the same body on the lineage in Z-order -- other summation orders in the batch-norm statistics and
    # the weight gradients.  Compared in fp32 only (16-bit slabs round differently after the first batch norm and the
    # gradients of this random-initialised stack of ~30 train-mode batch norms amplify that beyond any useful bound; DESIGN.md
    # section 1, configs[2]).  The forward agrees to fp32 rounding; each path's gradients are bounded against the CPU oracle on
    # their own (tests/test_train_gpu.py: 3e-4 in the caller's row order, 8e-4 in Z-order, tolerance 2e-3).
    if zord is None:
        return

    def near(a, b, tol, what):
        a, b = a.float(), b.float()
        assert (a - b).norm().item() <= tol * max(b.norm().item(), 1e-12), what

    near(zord[0], ref[0], 1e-5, "output")
    if want_dx:
        near(zord[1], ref[1], 5e-3, "input gradient")
    for k in ref[2]:
        near(zord[2][k], ref[2][k], 5e-3, k)
    for k in ref[3]:
        near(zord[3][k], ref[3][k], 1e-4, k)


def test_engine_two_steps_with_optimizer():
    """Weights change between steps (packed forms are refreshed), the plan is reused, losses follow the module path."""
    coords = _coords(seed=62, batch=1)
    feats = torch.randn(len(coords), 6, device=DEV).to(torch.bfloat16)
    target = torch.randn(len(coords), 32, device=DEV)
    losses = {}
    try:
        TE.SORTED = False
        for engine in (False, True):
            TE.ENABLED = engine
            torch.manual_seed(9)
            net = Mink_unet(6, 32, arch="MinkUNet14A").to(DEV).train()
            opt = torch.optim.Adam(net.parameters(), lr=1e-2)
            ls = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                out = net(ME.SparseTensor(feats, torch.from_numpy(coords).to(DEV))).F
                loss = ((out.float() - target) ** 2).mean()
                loss.backward()
                opt.step()
                ls.append(loss.item())
            losses[engine] = ls
    finally:
        TE.ENABLED, TE.SORTED = True, False
    assert losses[True][0] == losses[False][0], losses              # same forward
    for a, b in zip(losses[True], losses[False]):                    # 16-bit gradients agree to a few roundings (see above)
        assert abs(a - b) <= 5e-3 * abs(b), losses
    assert losses[True][2] < losses[True][0]


def test_engine_eval_and_no_grad_fall_back():
    coords = _coords(seed=63, batch=1)
    net = Mink_unet(6, 32, arch="MinkUNet14A").to(DEV).train()
    feats = torch.randn(len(coords), 6, device=DEV).to(torch.bfloat16)
    with torch.no_grad():
        net(ME.SparseTensor(feats, torch.from_numpy(coords).to(DEV)))
    assert not net.__dict__.get("_train_plans")
    net.block3[0].norm1.bn.eval()                        # one frozen batch norm: the executor would normalise with batch statistics
    out = net(ME.SparseTensor(feats, torch.from_numpy(coords).to(DEV))).F
    out.float().sum().backward()
    assert not net.__dict__.get("_train_plans")
    net.train()
    net(ME.SparseTensor(feats, torch.from_numpy(coords).to(DEV))).F.float().sum().backward()
    assert net.__dict__.get("_train_plans")


@pytest.mark.parametrize("n_points", [40, 700])
def test_engine_on_tiny_scenes(n_points):
    """A handful of voxels: the coarse levels hold one to a few rows (batch-norm statistics over one row, weight-gradient
    workgroups without a single pair, quarter tiles): fp32, bit-identical to the module path."""
    g = torch.Generator().manual_seed(n_points)
    xyz = torch.randint(0, 24 if n_points < 100 else 40, (n_points, 3), generator=g, dtype=torch.int32)
    coords = torch.unique(torch.cat([torch.zeros(n_points, 1, dtype=torch.int32), xyz], 1), dim=0).numpy()
    torch.manual_seed(3)
    net = Mink_unet(6, 32, arch="MinkUNet14A").to(DEV).train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    feats = torch.randn(len(coords), 6, device=DEV)
    target = torch.randn(len(coords), 32, device=DEV)
    try:
        ref = _step(net, feats, coords, target, False, True)
        net.load_state_dict(state)
        got = _step(net, feats, coords, target, True, True)
        assert net.__dict__.get("_train_plans"), "the executor did not run"
    finally:
        TE.ENABLED, TE.SORTED = True, False
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    for k in ref[2]:
        assert torch.equal(got[2][k], ref[2][k]), k
        assert torch.isfinite(got[2][k]).all(), k


def test_engine_refuses_what_autograd_cannot_see():
    """The backward re-reads the activation arena and the packed kernels of the forward: a second backward through the same
    forward, a parameter changed in between and (for the node's own output) an in-place modification are refused loudly."""
    coords = _coords()
    torch.manual_seed(7)
    net = Mink_unet(6, 32, arch="MinkUNet14A").to(DEV).train()
    feats = torch.randn(len(coords), 6, device=DEV)
    TE.ENABLED, TE.SORTED = True, False

    def fwd():
        return net(ME.SparseTensor(feats, torch.from_numpy(coords).to(DEV))).F
    out = fwd()
    assert net.__dict__.get("_train_plans"), "the executor did not run"
    out.float().sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="backward ran twice"):
        out.float().sum().backward()
    out = fwd()
    with torch.no_grad():
        net.conv0p1s1.kernel.add_(1e-3)            # what an optimizer step between forward and backward would do
    with pytest.raises(RuntimeError, match="parameter changed"):
        out.float().sum().backward()
    for p in net.parameters():
        p.grad = None
    out = fwd()
    out.float().sum().backward()                   # and the ordinary order still works
    assert net.conv0p1s1.kernel.grad is not None
