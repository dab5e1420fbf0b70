"""GPU tier, training path (BASELINE configs[2]): gradients of the sparse convolutions (dgrad on the implicit-GEMM
kernel, wgrad on the matrix cores over the rule pairs) against torch autograd through the CPU oracle's gather-mm-index_add convolution."""
import numpy as np
import pytest
import torch

from oracle import sparse_ref as R
import pbnet_amd.MinkowskiEngine as ME
from pbnet_amd import synth
from pbnet_amd.config import get_config
from pbnet_amd.network.Mink import Mink_unet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _coords(seed=51, batch=2):
    sc = synth.synth_room(seed=seed, pitch=0.0225, room=(0.6, 0.5, 0.4), n_boxes=1)
    q, _, _ = synth.voxelize_numpy(sc["xyz"], 0.02)
    parts = [np.concatenate([np.full((len(q), 1), b, np.int32), q + np.array([7 * b, 0, 0], np.int32)], 1) for b in range(batch)]
    return np.concatenate(parts, 0).astype(np.int32)


def _rel(a, b):
    return (a - b).norm().item() / max(b.norm().item(), 1e-12)


@pytest.mark.parametrize("kind", ["k3", "k5", "down", "up", "1x1", "linear"])
def test_convolution_gradients(kind):
    coords = _coords()
    cm_ref = R.CoordinateManager(coords)
    torch.manual_seed(7)
    cin, cout = (6, 32) if kind == "k5" else (32, 48)
    n1, n2 = len(coords), cm_ref.get_coords(2).shape[0]
    if kind in ("k3", "k5", "1x1"):
        k = {"k3": 3, "k5": 5, "1x1": 1}[kind]
        mod = ME.MinkowskiConvolution(cin, cout, kernel_size=k, bias=(kind == "1x1"), dimension=3)
        n_in, n_out, in_stride = n1, n1, 1
        ref = lambda x, w, b: R.conv(x, w, None if k == 1 else cm_ref.get_map(1, 1, k), n1, bias=b)
    elif kind == "down":
        mod = ME.MinkowskiConvolution(cin, cout, kernel_size=2, stride=2, dimension=3)
        n_in, n_out, in_stride = n1, n2, 1
        ref = lambda x, w, b: R.conv(x, w, cm_ref.get_map(1, 2, 2), n2)
    elif kind == "up":
        mod = ME.MinkowskiConvolutionTranspose(cin, cout, kernel_size=2, stride=2, dimension=3)
        n_in, n_out, in_stride = n2, n1, 2
        ref = lambda x, w, b: R.conv_transpose(x, w, cm_ref.get_map(1, 2, 2), n1)
    else:
        mod = ME.MinkowskiLinear(cin, cout, bias=True)
        n_in, n_out, in_stride = n1, n1, 1
        ref = None
    x0 = torch.randn(n_in, cin)
    gy = torch.randn(n_out, cout)
    # reference gradients (CPU autograd)
    xr = x0.clone().requires_grad_(True)
    if kind == "linear":
        wr = mod.linear.weight.detach().clone().requires_grad_(True)
        br = mod.linear.bias.detach().clone().requires_grad_(True)
        yr = xr @ wr.t() + br
    else:
        wr = mod.kernel.detach().clone().requires_grad_(True)
        br = mod.bias.detach().clone().requires_grad_(True) if mod.bias is not None else None
        yr = ref(xr, wr, br)
    (yr * gy).sum().backward()
    # device path
    mod = mod.to(DEV)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(DEV))
    xd = x0.to(DEV).requires_grad_(True)
    yd = mod(ME.SparseTensor(xd, coordinate_manager=cm, tensor_stride=in_stride)).F
    assert _rel(yd.detach().cpu(), yr.detach()) < 1e-5
    (yd * gy.to(DEV)).sum().backward()
    assert _rel(xd.grad.cpu(), xr.grad) < 1e-5, "input gradient"
    w_dev = mod.linear.weight if kind == "linear" else mod.kernel
    assert _rel(w_dev.grad.cpu(), wr.grad) < 1e-5, "weight gradient"
    if br is not None:
        b_dev = mod.linear.bias if kind == "linear" else mod.bias
        assert _rel(b_dev.grad.cpu().view(-1), br.grad.view(-1)) < 1e-5, "bias gradient"


@pytest.mark.parametrize("kind", ["k3", "k5", "down", "up", "1x1"])
def test_convolution_gradients_bf16(kind):
    """The same layer kinds with bf16 slabs (BASELINE configs[2]): forward, input gradient (dgrad on k_spconv) and
    weight gradient against fp32 autograd through the oracle ON THE bf16-ROUNDED OPERANDS, so that what is measured is
    the kernels' own error (fp32 accumulation of bf16 products, one rounding of the result), not the input rounding."""
    coords = _coords()
    cm_ref = R.CoordinateManager(coords)
    torch.manual_seed(8)
    cin, cout = (6, 32) if kind == "k5" else (32, 48)
    n1, n2 = len(coords), cm_ref.get_coords(2).shape[0]
    if kind in ("k3", "k5", "1x1"):
        k = {"k3": 3, "k5": 5, "1x1": 1}[kind]
        mod = ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3)
        n_in, n_out, in_stride = n1, n1, 1
        ref = lambda x, w: R.conv(x, w, None if k == 1 else cm_ref.get_map(1, 1, k), n1)
    elif kind == "down":
        mod = ME.MinkowskiConvolution(cin, cout, kernel_size=2, stride=2, dimension=3)
        n_in, n_out, in_stride = n1, n2, 1
        ref = lambda x, w: R.conv(x, w, cm_ref.get_map(1, 2, 2), n2)
    else:
        mod = ME.MinkowskiConvolutionTranspose(cin, cout, kernel_size=2, stride=2, dimension=3)
        n_in, n_out, in_stride = n2, n1, 2
        ref = lambda x, w: R.conv_transpose(x, w, cm_ref.get_map(1, 2, 2), n1)
    q = lambda t: t.to(torch.bfloat16).float()
    x0, gy = q(torch.randn(n_in, cin)), q(torch.randn(n_out, cout))
    xr = x0.clone().requires_grad_(True)
    wr = q(mod.kernel.detach()).clone().requires_grad_(True)
    yr = ref(xr, wr)
    (yr * gy).sum().backward()
    mod = mod.to(DEV)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(DEV))
    xd = x0.to(DEV).to(torch.bfloat16).requires_grad_(True)
    yd = mod(ME.SparseTensor(xd, coordinate_manager=cm, tensor_stride=in_stride)).F
    assert yd.dtype == torch.bfloat16
    r_y = _rel(yd.detach().float().cpu(), yr.detach())
    (yd * gy.to(DEV).to(torch.bfloat16)).sum().backward()
    r_x, r_w = _rel(xd.grad.float().cpu(), xr.grad), _rel(mod.kernel.grad.float().cpu(), wr.grad)
    print("%s bf16: forward rel %.2e, dgrad rel %.2e, wgrad rel %.2e" % (kind, r_y, r_x, r_w))
    assert r_y < 4e-3 and r_x < 4e-3 and r_w < 4e-3      # one bf16 rounding of the result: 2^-9 = 2e-3 per element


def test_unet_training_step_gradients():
    """MinkUNet14A in train mode: loss.backward() through every layer kind; gradients vs oracle autograd."""
    coords = _coords(seed=52)
    torch.manual_seed(22)
    net = Mink_unet(6, 32, arch="MinkUNet14A")
    feats = torch.randn(len(coords), 6)
    target = torch.randn(len(coords), 32)
    # oracle
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
          for k, v in net.state_dict().items()}
    xr = feats.clone().requires_grad_(True)
    out_r = R.minkunet_forward(sd, "MinkUNet14A", xr, coords, training=True, detach=False)
    loss_r = ((out_r - target) ** 2).mean()
    loss_r.backward()
    # device
    net = net.to(DEV).train()
    xd = feats.to(DEV).requires_grad_(True)
    out_d = net(ME.SparseTensor(xd, torch.from_numpy(coords).to(DEV))).F
    loss_d = ((out_d - target.to(DEV)) ** 2).mean()
    loss_d.backward()
    assert abs(loss_d.item() - loss_r.item()) <= 1e-4 * max(1.0, abs(loss_r.item()))
    assert _rel(xd.grad.cpu(), xr.grad) < 1e-3
    named = dict(net.named_parameters())
    checked = 0
    for name in ("conv0p1s1.kernel", "conv1p1s2.kernel", "block1.0.conv1.kernel", "block4.0.conv2.kernel",
                 "block5.0.downsample.0.kernel", "convtr4p16s2.kernel", "convtr7p2s2.kernel", "block8.0.conv2.kernel",
                 "final_sematic.kernel", "final_sematic.bias", "bn0.bn.weight", "block3.0.norm1.bn.bias"):
        g_dev, g_ref = named[name].grad, sd[name].grad
        assert g_dev is not None and g_ref is not None, name
        assert _rel(g_dev.cpu(), g_ref) < 2e-3, name
        checked += 1
    assert checked == 12
    assert all(p.grad is not None for p in net.parameters())


def test_pbnet_training_step_runs():
    """model_fn (PBNet.py:349-444) with the cluster stage on: forward, losses (incl. pbnet_ops.get_iou), backward."""
    from pbnet_amd.network.PBNet import PBNet, model_fn
    cfg = get_config(batch_size=2, cluster_epoch=0)
    torch.manual_seed(22)
    model = PBNet(cfg).to(DEV).train()
    batch_np, teacher_np, _ = synth.make_val_batch(seed=3, copies=2, room=(1.2, 1.0, 0.8), n_boxes=4, pitch=0.03,
                                                   classes=(17, 10))
    t = torch.from_numpy
    batch = {k: t(v) for k, v in batch_np.items()}
    n = batch["xyz_original"].shape[0]
    ins = batch["ins"]
    n_inst = int(ins.max().item()) + 1
    sem = teacher_np["sem_score"].argmax(1)
    info = torch.zeros(n, 9)
    pointnum = []
    for i in range(n_inst):
        m = ins == i
        pointnum.append(int(m.sum()))
        if m.any():
            info[m, 0:3] = batch["xyz_original"][m].mean(0)
    batch.update(sem=t(sem).long(), inst_info=info, instance_pointnum=torch.tensor(pointnum, dtype=torch.int32))
    # teacher-forced heads so that the grouping stage finds instances (random heads cannot): patch forward's hook
    teacher = {k: t(v) for k, v in teacher_np.items()}
    orig_forward = model.forward
    model.forward = lambda *a, **kw: orig_forward(*a, teacher=teacher, **kw)
    loss, pred, visual, meter = model_fn(batch, model, 1, cfg, "train")
    assert torch.isfinite(loss)
    assert "mask_loss" in visual and pred["proposals"][1].shape[0] > 1
    loss.backward()
    got = [n_ for n_, p in model.named_parameters() if p.grad is not None]
    # the mask / score branches learn through their own U-Nets; the backbone through the mask features
    for prefix in ("D_Unet.", "score_Unet.", "linear_binary.", "linear_IOU.", "MEUnet."):
        assert any(g.startswith(prefix) for g in got), prefix
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # round 4: the index-only glue between the networks on the fused launches (PBNet.TRAIN_FUSED_GLUE, the default) against the
    # plain-torch glue on the same weights and batch: the same integers (proposals), the same loss, the same gradients
    import pbnet_amd.network.PBNet as PB
    assert PB.TRAIN_FUSED_GLUE
    fused = (loss.detach().clone(), [x.clone() for x in pred["proposals"][:3]], pred["clt_scores"].detach().clone(),
             {n_: p.grad.detach().clone() for n_, p in model.named_parameters() if p.grad is not None})
    try:
        PB.TRAIN_FUSED_GLUE = False
        for p in model.parameters():
            p.grad = None
        batch2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        loss2, pred2, _, _ = model_fn(batch2, model, 1, cfg, "train")
        loss2.backward()
    finally:
        PB.TRAIN_FUSED_GLUE = True
    for a, b in zip(fused[1], pred2["proposals"][:3]):
        assert torch.equal(a.cpu(), b.cpu())
    # train-mode batch norm updated its running statistics between the two forwards: statistics of the batch, not the running ones,
    # normalise in training, so the outputs are the same numbers
    assert torch.equal(fused[2], pred2["clt_scores"].detach())
    assert abs(float(fused[0]) - float(loss2.detach())) <= 1e-6 * max(1.0, abs(float(loss2.detach())))
    grads2 = {n_: p.grad for n_, p in model.named_parameters() if p.grad is not None}
    assert set(grads2) == set(fused[3])
    for n_, g in grads2.items():
        d = (g.float() - fused[3][n_].float()).abs().max().item()
        assert d <= 1e-5 * max(1.0, g.float().abs().max().item()), (n_, d)


def test_rulebook_pairs_against_nonzero():
    """pbn_rulebook_pair_* (offset-major pair lists of the weight gradient) against torch.nonzero on random maps: same
    pairs in the same order inside every offset, padding = -1, segment -> offset table, surplus-free segment count."""
    from pbnet_amd.MinkowskiEngine.conv import rulebook_pairs
    g = torch.Generator().manual_seed(3)
    for v, k, density, seg in ((1000, 27, 0.3, 64), (70001, 27, 0.26, 4096), (333, 8, 0.12, 16), (5000, 125, 0.15, 512),
                               (257, 1, 1.0, 100), (40, 27, 0.0, 8)):
        nbr = torch.where(torch.rand(v, k, generator=g) < density, torch.randint(0, v, (v, k), generator=g),
                          torch.full((v, k), -1)).to(torch.int32).to(DEV)
        in_idx, out_idx, seg_offset, n_seg = rulebook_pairs(nbr, seg)
        cnt = (nbr >= 0).sum(0).cpu()
        assert n_seg == int(((cnt + seg - 1) // seg).sum()) and in_idx.shape[0] == n_seg * seg == out_idx.shape[0]
        kv = (nbr >= 0).t().nonzero().cpu()                               # offset-major, output rows ascending
        want_in = nbr.cpu()[kv[:, 1], kv[:, 0]].long()
        valid = (in_idx >= 0).cpu()
        assert torch.equal(valid, (out_idx >= 0).cpu()) and int(valid.sum()) == kv.shape[0]
        assert in_idx.dtype == torch.int32 and out_idx.dtype == torch.int32
        assert torch.equal(in_idx.cpu()[valid].long(), want_in) and torch.equal(out_idx.cpu()[valid].long(), kv[:, 1])
        seg_of_slot = torch.arange(n_seg * seg) // seg
        assert torch.equal(seg_offset.cpu()[seg_of_slot[valid]], kv[:, 0])
        # padding only at the tail of an offset's last segment
        per_seg = valid.view(n_seg, seg).sum(1) if n_seg else valid.sum()
        if n_seg:
            assert ((per_seg > 0).all()) and (valid.view(n_seg, seg).long().diff(dim=1) <= 0).all()


def test_rulebook_pairs_multi_equals_single():
    """pbn_rulebook_pairs_multi (all maps of a lineage in three launches) against the per-map device-side fill: the same
    lists entry for entry (below every offset's pair count), the same segment starts and counts; an empty map among them."""
    from pbnet_amd.MinkowskiEngine.conv import rulebook_pairs_dev, rulebook_pairs_dev_multi, WGRAD_PAIR_SEGMENT as SEG
    g = torch.Generator().manual_seed(11)
    shapes = [(3000, 27, 0.3), (70001, 27, 0.26), (333, 8, 0.12), (5000, 125, 0.15), (900, 8, 0.125), (257, 27, 0.0)]

    def make():
        gg = torch.Generator().manual_seed(11)
        return [torch.where(torch.rand(v, k, generator=gg) < d, torch.randint(0, v, (v, k), generator=gg),
                            torch.full((v, k), -1)).to(torch.int32).to(DEV) for v, k, d in shapes]
    single = [rulebook_pairs_dev(m) for m in make()]
    multi = rulebook_pairs_dev_multi(make())
    for (v, k, _), a, b in zip(shapes, single, multi):
        cnt = a[3].cpu()
        assert torch.equal(cnt, b[3].cpu()) and torch.equal(a[2].cpu(), b[2].cpu())
        seg_begin = a[2].cpu().long()
        for o in range(k):
            lo, n = int(seg_begin[o]) * SEG, int(cnt[o])
            assert torch.equal(a[0][lo:lo + n], b[0][lo:lo + n]) and torch.equal(a[1][lo:lo + n], b[1][lo:lo + n])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_native_batch_norm_matches_torch(dtype, tol):
    """csrc/bnorm.hip (train-mode statistics, normalisation, gradients) against torch.nn.BatchNorm1d on the same slab:
    outputs, running statistics (unbiased variance), dx / dweight / dbias; channel means far from zero."""
    import pbnet_amd.MinkowskiEngine as ME
    from pbnet_amd.MinkowskiEngine.nn import MinkowskiBatchNorm
    g = torch.Generator().manual_seed(7)
    for n, c in ((1000, 32), (146001, 96), (77, 256), (5, 384), (4097, 64), (1, 32)):
        x0 = (torch.randn(n, c, generator=g) * (torch.rand(c, generator=g) * 3 + 0.1) + torch.randn(c, generator=g) * 5).to(DEV)
        gy = torch.randn(n, c, generator=g).to(DEV)
        outs = []
        for native in (True, False):
            m = MinkowskiBatchNorm(c).to(DEV).train()
            with torch.no_grad():
                m.bn.weight.copy_(torch.linspace(0.5, 1.5, c))
                m.bn.bias.copy_(torch.linspace(-1, 1, c))
            m.NATIVE_TRAIN = native
            x = x0.to(dtype).requires_grad_(True)
            i = torch.arange(n, dtype=torch.int32)
            coords = torch.stack([torch.zeros_like(i), i % 1000, i // 1000, torch.zeros_like(i)], 1).to(DEV)
            if n == 1 and not native:
                outs.append(None)                       # torch refuses one value per channel in training mode
                continue
            y = m(ME.SparseTensor(x, coords)).F
            y.backward(gy.to(dtype))
            outs.append((y.detach().float(), x.grad.float(), m.bn.weight.grad.clone(), m.bn.bias.grad.clone(),
                         m.bn.running_mean.clone(), m.bn.running_var.clone(), int(m.bn.num_batches_tracked)))
        if outs[1] is None:
            assert torch.isfinite(outs[0][0]).all()
            continue
        for k, (a, b) in enumerate(zip(outs[0], outs[1])):
            if k == 6:
                assert a == b == 1
                continue
            scale = max(1.0, float(b.abs().max()))
            t = tol if k < 2 else max(2e-5, tol * 1e-1) * (n if k in (2, 3) else 1) ** 0.5   # sums over n rows
            assert float((a - b).abs().max()) <= t * scale, (n, c, k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (True, False)])
def test_fused_batch_norm_tail_matches_the_separate_modules(dtype, tol, with_res, relu):
    """bn_act (csrc/bnorm.hip pbn_bn_act_train_*): relu(bn(x) [+ residual]) in the normalisation's own passes against
    norm -> += residual -> relu as separate modules: output, dx, d(residual), dweight, dbias, running statistics, and the
    lazily counted num_batches_tracked."""
    from pbnet_amd.MinkowskiEngine.nn import MinkowskiBatchNorm, bn_act
    g = torch.Generator().manual_seed(11)
    for n, c in ((1000, 32), (40001, 96), (77, 256), (3, 64)):
        x0 = (torch.randn(n, c, generator=g) * (torch.rand(c, generator=g) * 2 + 0.2) + torch.randn(c, generator=g)).to(DEV)
        r0 = torch.randn(n, c, generator=g).to(DEV)
        gy = torch.randn(n, c, generator=g).to(DEV)
        i = torch.arange(n, dtype=torch.int32)
        coords = torch.stack([torch.zeros_like(i), i % 1000, i // 1000, torch.zeros_like(i)], 1).to(DEV)
        outs = []
        for fused in (True, False):
            m = MinkowskiBatchNorm(c).to(DEV).train()
            with torch.no_grad():
                m.bn.weight.copy_(torch.linspace(0.5, 1.5, c))
                m.bn.bias.copy_(torch.linspace(-1, 1, c))
            m.FUSE_ACT = fused
            x = x0.to(dtype).requires_grad_(True)
            r = r0.to(dtype).requires_grad_(True) if with_res else None
            y = bn_act(m, ME.SparseTensor(x, coords), residual=None if r is None else ME.SparseTensor(r, coords), relu=relu).F
            y.backward(gy.to(dtype))
            outs.append((y.detach().float(), x.grad.float(), None if r is None else r.grad.float(), m.bn.weight.grad.clone(),
                         m.bn.bias.grad.clone(), m.bn.running_mean.clone(), m.bn.running_var.clone(),
                         int(m.state_dict()["bn.num_batches_tracked"]), int(m.bn.num_batches_tracked)))
        for k, (a, b) in enumerate(zip(outs[0], outs[1])):
            if a is None:
                assert b is None
                continue
            if k >= 7:
                assert a == b == 1
                continue
            scale = max(1.0, float(b.abs().max()))
            t = tol if k < 3 else max(2e-5, tol * 1e-1) * (n if k in (3, 4) else 1) ** 0.5
            if dtype != torch.float32 and relu and k in (1, 2, 3, 4):
                # the fused pass decides y > 0 on the fp32 value, the separate modules on the twice-rounded bf16 one: a
                # handful of outputs within one bf16 ulp of zero flip their mask, and with it their whole gradient
                bad = ((a - b).abs() > t * scale).float().mean().item() if k in (1, 2) else 0.0
                assert bad <= 2e-3, (n, c, k, bad)
                if k in (3, 4):
                    assert float((a - b).abs().max()) <= 10 * t * scale, (n, c, k, float((a - b).abs().max()), scale)
                continue
            assert float((a - b).abs().max()) <= t * scale, (n, c, k, float((a - b).abs().max()), scale)
        if relu:
            assert float(outs[0][0].min()) >= 0.0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-2)])
def test_fused_training_nodes_match_the_module_path(dtype, tol):
    """pbnet_amd/MinkowskiEngine/fused_train.py (conv -> bn -> relu and the whole residual block as ONE autograd node, the
    residual adds inside the convolution / normalisation epilogues) against the module-by-module path on the same
    weights: output, input gradient, every parameter gradient, running statistics."""
    from pbnet_amd.MinkowskiEngine import fused_train
    coords = _coords(seed=53)
    feats = torch.randn(len(coords), 6)
    target = torch.randn(len(coords), 32)
    res = []
    for fused in (True, False):
        torch.manual_seed(22)
        net = Mink_unet(6, 32, arch="MinkUNet18A").to(DEV).train()
        fused_train.ENABLED = fused
        try:
            xd = feats.to(DEV).to(dtype).requires_grad_(True)
            out = net(ME.SparseTensor(xd, torch.from_numpy(coords).to(DEV))).F
            loss = ((out.float() - target.to(DEV)) ** 2).mean()
            loss.backward()
        finally:
            fused_train.ENABLED = True
        res.append((out.detach().float(), xd.grad.float(), {k: p.grad.float() for k, p in net.named_parameters()},
                    {k: b.clone().float() for k, b in net.named_buffers()}))
    (o1, g1, p1, b1), (o0, g0, p0, b0) = res
    assert _rel(o1, o0) < tol and _rel(g1, g0) < tol
    worst = max((_rel(p1[k], p0[k]), k) for k in p0)
    assert worst[0] < tol * (1 if dtype == torch.float32 else 3), worst
    for k in b0:
        assert torch.allclose(b1[k], b0[k], rtol=1e-3 if dtype != torch.float32 else 1e-5, atol=1e-3 if dtype != torch.float32 else 1e-6), k


def test_segment_pool_with_autograd_matches_the_torch_reductions():
    """global max + avg pooling per proposal (PBNet.py:274-276) through pbn_segment_pool with the reductions' backward rules
    against MinkowskiGlobalMaxPooling + MinkowskiGlobalAvgPooling (scatter_reduce amax / index_add), rows grouped by id."""
    from pbnet_amd.MinkowskiEngine.nn import global_max_plus_avg_pool
    g = torch.Generator().manual_seed(3)
    sizes = [1, 700, 33, 5000, 2, 64]
    b = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    n = int(b.numel())
    coords = torch.stack([b.int(), torch.arange(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32)], 1).to(DEV)
    f0 = torch.randn(n, 16, generator=g)
    f0[10:20] = f0[10]                                    # ties inside a segment share the maximum's gradient
    gy = torch.randn(len(sizes), 16, generator=g).to(DEV)
    outs = []
    for native in (True, False):
        f = f0.clone().to(DEV).requires_grad_(True)
        x = ME.SparseTensor(f, coords)
        y = global_max_plus_avg_pool(x).F if native else (ME.MinkowskiGlobalMaxPooling()(x) + ME.MinkowskiGlobalAvgPooling()(x)).F
        y.backward(gy)
        outs.append((y.detach(), f.grad))
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-5
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-5
