"""One process per GPU.  Inference shards SCENES (independent units: /root/reference/eval_map.py:48-50 handles one
scene at a time) with no data-path collective; training adds exactly one exchange step per iteration, the gradient
all-reduce of /root/reference/train.py:345 (DDP), here as flat fp32 buckets sized for RCCL over xGMI.
SyncBatchNorm (train.py:343-344) is deliberately not reproduced (BASELINE.json north_star: "gradients only")."""
import torch
import torch.distributed as dist


def shard_scenes(n_scenes, rank, world):
    """Round-robin scene indices of this rank -- what DistributedSampler(shuffle=False) yields
    (/root/reference/datasets/scannetv2/dataset_preprocess.py:50,59), including its wrap-around padding so every rank
    runs the same number of steps."""
    per_rank = (n_scenes + world - 1) // world
    return [(rank + i * world) % n_scenes for i in range(per_rank)] if n_scenes > 0 else []


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scene_results(local, dst=0):
    """The one optional exchange of the inference path: every rank's per-scene records -- {scene name:
    record}, e.g. evaluate.SceneMatches -- merged on rank `dst` so that the AP of the whole validation set is computed
    once (eval_map.py:149-151 does it in its single process).  A scene that two ranks evaluated (the wrap-around padding
    of shard_scenes) is kept once, from the lower rank.  Returns the merged dict on `dst`, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local)
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = [None] * world if rank == dst else None
    dist.gather_object(dict(local), parts, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for part in parts:
        for scene, rec in part.items():
            merged.setdefault(scene, rec)
    return merged


def allreduce_gradients(params, bucket_bytes=64 << 20):
    """Average gradients across ranks in flat buckets (>= 64 MB keeps a ring on 7 x 153 GB/s xGMI links bandwidth-
    rather than latency-bound).  Parameters without a gradient on this rank (the mask/score
    branches while epoch <= cluster_epoch, train.py:345 find_unused_parameters=True) contribute zeros."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    n_buckets, i = 0, 0
    while i < len(params):
        bucket, size = [], 0
        while i < len(params) and (not bucket or size + params[i].numel() * 4 <= bucket_bytes):
            bucket.append(params[i])
            size += params[i].numel() * 4
            i += 1
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().float().reshape(-1)
                          for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        off = 0
        for p in bucket:
            g = flat[off:off + p.numel()].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += p.numel()
        n_buckets += 1
    return n_buckets
