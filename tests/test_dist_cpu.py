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
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    x = torch.full((3, 8), float(rank + 1))
    net[1](net[0](x)).sum().backward()            # net[2] gets no gradient (unused branch)
    local = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
    nb = pd.allreduce_gradients(net.parameters(), bucket_bytes=256)
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.tolist() if g is not None else None for g in local])
    ok = nb > 1
    for i, p in enumerate(net.parameters()):
        if gathered[0][i] is None:
            ok = ok and float(p.grad.abs().sum()) == 0.0
        else:
            want = sum(torch.tensor(gathered[r][i]) for r in range(world)) / world
            ok = ok and torch.allclose(p.grad, want, atol=1e-6)
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
