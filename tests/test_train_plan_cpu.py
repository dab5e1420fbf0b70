"""CPU tier: the static plan of the native training executor (network/train_engine.py): op list, in-place skip slabs, which
input gradients are accumulated, parameter-gradient layout.  (The plan is host logic; the executor itself is GPU tier:
tests/test_train_engine_gpu.py.)"""
import pytest
import torch

from pbnet_amd.network import train_engine as TE
from pbnet_amd.network.Mink import Mink_unet


@pytest.mark.parametrize("arch", ["MinkUNet14A", "MinkUNet34C"])
def test_plan_structure(arch):
    torch.manual_seed(0)
    net = Mink_unet(6, 20, arch=arch)
    plan = TE.TrainPlan(net, torch.bfloat16, want_input_grad=False)
    n_blocks = sum(net.LAYERS)
    n_short = sum(1 for i in range(1, 9) if getattr(net, "block%d" % i)[0].downsample is not None)
    assert len(plan.recs) == 1 + 8 + 2 * n_blocks + n_short
    # every body parameter exactly once, in (kernel, gamma, beta) triples; the final 1x1 stays outside
    body = [p for k, p in net.named_parameters() if not k.startswith("final_sematic")]
    assert len(plan.params) == len(body) and {id(p) for p in plan.params} == {id(p) for p in body}
    assert plan.grad_floats == sum(p.numel() for p in body) == sum(plan.split_sizes)
    ops = plan.ops
    # stem: buffer 0, no input gradient asked for
    assert ops[0].in_buf == 0 and ops[0].map_kind == 2 and ops[0].want_dx == 0
    kinds = [ops[i].map_kind for i in range(len(plan.recs))]
    assert kinds.count(3) == 4 and kinds.count(4) == 4 and kinds.count(2) == 1
    # skip slabs: the stem and the first three encoder stages write into the right-hand columns of a decoder slab, the
    # transposed convolutions into the left-hand columns, and the decoder's first block reads the whole slab
    slabs = {}
    for i in range(len(plan.recs)):
        o = ops[i]
        if o.map_kind == 4:
            assert o.out_col == 0
            slabs[o.out_buf] = plan.bufs[o.out_buf][1]
    assert len(slabs) == 4
    for i in range(len(plan.recs)):
        o = ops[i]
        if o.out_buf in slabs and o.map_kind != 4:
            assert o.out_col > 0 and o.out_col + o.cout == slabs[o.out_buf]
        if o.in_buf in slabs and o.in_col == 0:
            assert o.cin == slabs[o.in_buf]                         # decoder block: conv1 and the 1x1 shortcut
    # accumulation: of the consumers of one gradient view exactly the LAST one in forward order writes, the others add
    by_view = {}
    for i in range(len(plan.recs)):
        o = ops[i]
        if o.want_dx:
            by_view.setdefault((o.in_buf, o.in_col if o.in_buf not in slabs else "slab"), []).append((i, o.dx_accumulate))
    for view, uses in by_view.items():
        accs = [a for _, a in uses]
        res_written = any(ops[j].res_buf == view[0] and view[1] != "slab" and ops[j].res_col == view[1] for j in range(len(plan.recs)))
        if view[1] == "slab":
            assert accs[-1] == 0 and all(a == 1 for a in accs[:-1]), (view, uses)
        elif res_written:
            assert all(a == 1 for a in accs), (view, uses)          # the block's batch-norm backward wrote the residual gradient first
        else:
            assert accs[-1] == 0 and all(a == 1 for a in accs[:-1]), (view, uses)
    with_dx = TE.TrainPlan(net, torch.bfloat16, want_input_grad=True)
    assert with_dx.ops[0].want_dx == 1 and with_dx.ops[0].dx_accumulate == 0 and with_dx.dinput_width == 16


def test_plan_rejects_partial_views():
    cov = TE._Covered()
    cov.mark(3, 0, 64)
    assert cov.state(3, 0, 64) == "full" and cov.state(3, 64, 128) == "none" and cov.state(4, 0, 8) == "none"
    with pytest.raises(NotImplementedError):
        cov.state(3, 32, 96)
    cov.mark(3, 64, 128)
    assert cov.state(3, 0, 128) == "full" and cov.state(3, 96, 128) == "full"
