"""CPU tier: the N>1 path (scene sharding, max-over-ranks timing, gradient all-reduce) on gloo, world_size 2."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pbnet_amd import dist as pd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)

    def make():
        torch.manual_seed(0)
        return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2),
                                   torch.nn.BatchNorm1d(4))
    x = torch.full((3, 8), float(rank + 1))
    # (1) post-backward form; net[2] is unused on every rank, net[1] only on rank 1 (contributes zeros there)
    net = make()
    h = net[0](x)
    (net[1](h).sum() if rank == 0 else h.sum()).backward()
    local = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
    nb = pd.allreduce_gradients(net.parameters(), bucket_bytes=256)
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.tolist() if g is not None else None for g in local])
    ok = nb > 1
    names = [n for n, _ in net.named_parameters()]
    for i, p in enumerate(net.parameters()):
        have = [gathered[r][i] for r in range(world) if gathered[r][i] is not None]
        if not have:                                   # unused on all ranks: grad stays None (DDP find_unused semantics)
            ok = ok and p.grad is None
        else:
            want = sum(torch.tensor(g) for g in have) / world
            ok = ok and p.grad is not None and torch.allclose(p.grad, want, atol=1e-6)
    ok = ok and net[2].weight.grad is None and net[1].weight.grad is not None
    # an optimiser step must not create state for the never-trained weights
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.step()
    ok = ok and net[2].weight not in opt.state and net[0].weight in opt.state
    # (2) overlapped form: hooks issue the buckets during backward; same numbers as (1), twice in a row
    net2 = make()
    red = pd.GradientReducer(net2.parameters(), bucket_bytes=256)
    for _ in range(2):
        for p in net2.parameters():
            p.grad = None
        h = net2[0](x)
        (net2[1](h).sum() if rank == 0 else h.sum()).backward()
        nb2 = red.finish()
        for pa, pb in zip(net.parameters(), net2.parameters()):
            ok = ok and ((pa.grad is None) == (pb.grad is None)) and (pa.grad is None or torch.allclose(pa.grad, pb.grad, atol=1e-6))
    ok = ok and nb2 == nb
    # (3) bf16 on the wire
    net3 = make()
    h = net3[0](x)
    (net3[1](h).sum() if rank == 0 else h.sum()).backward()
    pd.allreduce_gradients(net3.parameters(), bucket_bytes=256, comm_dtype=torch.bfloat16)
    for pa, pb in zip(net.parameters(), net3.parameters()):
        ok = ok and (pa.grad is None or torch.allclose(pa.grad, pb.grad, rtol=2e-2, atol=1e-2))
    # (3b) gradients that already live in the bucket: finish() hands out views of the flat buffer; zero_grad(set_to_none=False)
    # keeps them, two backwards accumulate into them in place (overlap=False), the second round packs nothing onto itself
    net4 = make()
    red4 = pd.GradientReducer(net4.parameters(), bucket_bytes=256, overlap=False)
    for rnd in range(2):
        for p in net4.parameters():
            if p.grad is not None:
                p.grad.zero_()
        for _ in range(2):
            h = net4[0](x)
            (net4[1](h).sum() if rank == 0 else h.sum()).backward()
        red4.finish()
        for pa, pb in zip(net.parameters(), net4.parameters()):
            ok = ok and ((pa.grad is None) == (pb.grad is None)) and (pa.grad is None or torch.allclose(2 * pa.grad, pb.grad, atol=1e-6))
        views = [b["views"][i].data_ptr() for b in red4.buckets for i in range(len(b["params"]))]
        ok = ok and all(p.grad is None or p.grad.data_ptr() in views for p in net4.parameters())
    # (4) BatchNorm statistics: every rank ends with rank 0's buffers
    with torch.no_grad():
        net[3].running_mean.fill_(float(rank + 1))
        net[3].num_batches_tracked.fill_(rank + 5)
    n_buf = pd.sync_buffers(net)
    ok = ok and n_buf == 3 and float(net[3].running_mean[0]) == 1.0 and int(net[3].num_batches_tracked) == 5
    t = pd.max_over_ranks(1.0 + rank)
    out[rank] = (ok, t, pd.shard_scenes(5, rank, world))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world))
    assert out[0][1] == out[1][1] == 2.0
    shards = [out[r][2] for r in range(world)]
    assert len(shards[0]) == len(shards[1]) == 3
    assert set(shards[0]) | set(shards[1]) == set(range(5))


def test_shard_single_rank_is_identity():
    assert pd.shard_scenes(4, 0, 1) == [0, 1, 2, 3]
    assert pd.shard_scenes(0, 0, 2) == []
    assert pd.max_over_ranks(3.5) == 3.5


def _eval_worker(rank, world, port, out):
    """Sharded validation: each rank associates its scenes, rank 0 merges and computes the AP of the whole set."""
    import numpy as np
    from pbnet_amd import evaluate as E
    from tests.test_oracle_eval import CASES, _record
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(CASES[1])                                           # E2: three scenes
    mine = pd.shard_scenes(int(g["n_scenes"]), rank, world)         # rank 0: [0, 2], rank 1: [1, 0] (wrap-around)
    local = {}
    for s in mine:
        rec = _record(g, s)
        if rank == 1 and s == 0:
            rec.pred_conf = rec.pred_conf * 0                       # the duplicate must lose against rank 0's copy
        local[rec.scene] = rec
    merged = pd.gather_scene_results(local)
    if rank == 0:
        ap = E.evaluate_matches(merged)
        same = np.array_equal(np.isnan(ap), np.isnan(g["ap"])) and np.array_equal(np.nan_to_num(ap), np.nan_to_num(g["ap"]))
        out[0] = (sorted(merged) == ["scene%04d_00" % s for s in range(3)], bool(same))
    else:
        out[rank] = (merged is None, True)
    dist.destroy_process_group()


def test_sharded_validation_merges_on_rank0():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_eval_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0] == (True, True) and out[1] == (True, True)
    assert pd.gather_scene_results({"a": 1}) == {"a": 1}           # no process group: identity
