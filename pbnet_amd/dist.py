"""One process per GPU.  Inference shards SCENES (independent units: /root/reference/eval_map.py:48-50 handles one
scene at a time) with no data-path collective; training adds exactly one exchange step per iteration, the gradient
all-reduce of /root/reference/train.py:345 (DDP), here as flat buckets sized for RCCL over xGMI and issued from
gradient hooks so that the ring overlaps backward (GradientReducer).  SyncBatchNorm (train.py:343-344) is deliberately
not reproduced (BASELINE.json north_star: "gradients only"); sync_buffers() makes the BatchNorm statistics of all ranks
equal before validation / checkpointing."""
import os

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


class GradientReducer(object):
    """The one exchange step of training (/root/reference/train.py:345, DistributedDataParallel with
    find_unused_parameters=True), re-designed for RCCL over xGMI rather than translated:

      * parameters are cut into flat buckets in REVERSE registration order (the order backward produces gradients);
        a bucket's all-reduce is issued asynchronously from the post-accumulate hook of its last gradient, so the ring
        runs beside the rest of backward (on `nccl` = RCCL the collective runs on the process group's own stream);
      * buckets are >= 64 MB by default: xGMI is point to point (7 links x ~153 GB/s per GPU), a ring step is per-link
        bound, and few large messages keep it bandwidth- rather than latency-bound (84 M parameters: 336 MB fp32,
        168 MB with comm_dtype=torch.bfloat16);
      * a parameter that received no gradient on ANY rank keeps grad None, exactly like DDP with
        find_unused_parameters=True (the mask / score branches while epoch <= cluster_epoch): a per-parameter `used`
        flag travels with a MAX reduction, so Adam creates no state and applies no decay for untrained weights; a
        parameter unused on this rank but used elsewhere contributes zeros.

    Usage:  r = GradientReducer(model.parameters());  loss.backward();  r.finish();  optimizer.step()"""

    def __init__(self, params, bucket_bytes=64 << 20, comm_dtype=torch.float32, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        self.comm_dtype = comm_dtype
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        esz = torch.empty(0, dtype=comm_dtype).element_size()
        self.buckets = []                       # each: dict(params, offsets, flat, pending, work)
        cur, size = [], 0
        for p in reversed(self.params):
            if cur and size + p.numel() * esz > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel() * esz
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        packed = []
        for bi, plist in enumerate(self.buckets):
            offs, off = [], 0
            for p in plist:
                offs.append(off)
                off += p.numel()
                self._bucket_of[id(p)] = (bi, len(offs) - 1)
            packed.append(dict(params=plist, offsets=offs, numel=off, flat=None, views=None, pending=len(plist), work=None,
                               filled=[False] * len(plist)))
        self.buckets = packed
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._next = 0                          # next bucket index to go on the wire
        self._hooks = []
        # which parameters received a gradient is HOST knowledge (p.grad is None or not): the ranks exchange it over a host-side
        # (gloo) group -- round 2 sent it through the device and read it back, one pipeline drain per step.  Collective: every
        # rank constructs its reducer at the same point (as it must for the buckets to match).
        self._host_group = None
        if self.world > 1 and dist.get_backend() != "gloo" and os.environ.get("PBN_REDUCER_HOST_GROUP", "1") != "0":
            self._host_group = _shared_host_group()
        if overlap and self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def wire_stats(self):
        """(buckets, bytes each rank puts on the wire per step, parameters): what a step's all-reduces move."""
        esz = torch.empty(0, dtype=self.comm_dtype).element_size()
        return {"buckets": len(self.buckets), "bytes_per_step": int(sum(b["numel"] for b in self.buckets) * esz),
                "parameters": len(self.params), "largest_bucket_bytes": int(max([b["numel"] for b in self.buckets] + [0]) * esz),
                "host_group_for_used_flags": self._host_group is not None}

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def reset(self):
        """Forget the gradients of a backward whose step is being SKIPPED (a non-finite loss followed by `continue`): without
        it the next backward would find its buckets filled and raise.  Every rank must skip the same steps -- the collectives
        of a step are issued in finish() (and from the hooks) on all ranks or on none: buckets that already went on the wire are
        waited for here so that the next step starts with a clean queue."""
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
            b["work"], b["pending"], b["filled"] = None, len(b["params"]), [False] * len(b["params"])
        self._next = 0

    # ---- per-gradient path (called by autograd during backward) ----------------------------------------------------
    def _flat(self, b, like):
        if b["flat"] is None or b["flat"].device != like.device:
            b["flat"] = torch.empty(b["numel"], dtype=self.comm_dtype, device=like.device)
            b["views"] = [b["flat"][o:o + p.numel()].view_as(p) for o, p in zip(b["offsets"], b["params"])]
        return b["flat"]

    def _pack(self, b, like):
        """Gradients of one bucket -> its flat buffer: ONE multi-tensor copy for the parameters that have a gradient on this
        rank, one multi-tensor zero for those that do not (round 3 issued a copy per parameter from the hook: ~470 tiny
        launches per step and as many again on the way back).  A gradient that already lives in the bucket (the views
        finish() hands out, kept by zero_grad(set_to_none=False) or by gradient accumulation) is not copied onto itself."""
        self._flat(b, like)
        dst, src, zero = [], [], []
        for v, p in zip(b["views"], b["params"]):
            g = p.grad
            if g is None:
                zero.append(v)
            elif g.data_ptr() != v.data_ptr() or g.dtype != v.dtype or not g.is_contiguous():
                dst.append(v)
                src.append(g.detach())
        if dst:
            torch._foreach_copy_(dst, src)
        if zero:
            torch._foreach_zero_(zero)

    def _on_grad(self, p):
        bi, pi = self._bucket_of[id(p)]
        b = self.buckets[bi]
        if b["filled"][pi]:
            # a second backward before finish(): the bucket may already have sent the FIRST gradient only, the
            # accumulated one would be lost -- refuse instead of averaging the wrong thing
            raise RuntimeError("GradientReducer: backward ran twice before finish(); with gradient accumulation build the "
                               "reducer with overlap=False (the all-reduce then starts in finish())")
        b["filled"][pi] = True
        b["pending"] -= 1
        # every rank must issue the same collectives in the same order: buckets go out strictly by index (a bucket
        # that completes early waits for its predecessors; what backward never completes is issued by finish())
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            nb = self.buckets[self._next]
            self._pack(nb, p)
            nb["work"] = dist.all_reduce(nb["flat"], op=dist.ReduceOp.SUM, async_op=True)
            self._next += 1

    # ---- after backward -------------------------------------------------------------------------------------------
    def finish(self):
        """Issue what backward did not, wait, hand the averaged gradients back.  Returns the number of buckets.
        After finish() `p.grad` is a VIEW of its bucket (fp32 on the wire; a converted copy of the bucket otherwise): no
        copy back, the optimiser reads the bucket.  The bucket is rewritten by the next step's backward."""
        if self.world == 1:
            return 0
        dev = next((p.grad.device for p in self.params if p.grad is not None), self.params[0].device)
        on_host = self._host_group is not None or dist.get_backend() == "gloo"
        used = torch.tensor([1 if p.grad is not None else 0 for p in self.params], dtype=torch.int32,
                            device="cpu" if on_host else dev)
        like = torch.empty(0, device=dev)
        for b in self.buckets[self._next:]:
            self._pack(b, like)
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)
        used_work = dist.all_reduce(used, op=dist.ReduceOp.MAX, async_op=True, group=self._host_group)
        self._next = 0
        used_work.wait()
        used_h = used.tolist()                   # a device tensor (no host group) is read back here: one synchronisation
        for b in self.buckets:
            b["work"].wait()
            flat = b["flat"]
            flat.div_(self.world)                # one launch per bucket
            typed = {}
            for pi, p in enumerate(b["params"]):
                if not used_h[self._index[id(p)]]:
                    p.grad = None                # unused on every rank: the optimiser must skip it (DDP semantics)
                    continue
                if p.dtype == flat.dtype:
                    p.grad = b["views"][pi]
                else:
                    if p.dtype not in typed:
                        typed[p.dtype] = flat.to(p.dtype)         # one conversion per bucket and dtype
                    o = b["offsets"][pi]
                    p.grad = typed[p.dtype][o:o + p.numel()].view_as(p)
            b["work"], b["pending"], b["filled"] = None, len(b["params"]), [False] * len(b["params"])
        return len(self.buckets)


_HOST_GROUP = {"made": False, "group": None}


def _shared_host_group():
    """ONE host-side (gloo) group per process, created at the first reducer and kept (round 3 made one per reducer and never
    destroyed it).  Whether it exists is agreed COLLECTIVELY: a rank whose gloo group failed must not take the device path
    while the others use the host group -- the all-reduces would mismatch and hang."""
    if _HOST_GROUP["made"]:
        return _HOST_GROUP["group"]
    grp, ok = None, 1
    try:
        grp = dist.new_group(backend="gloo")
    except Exception:                           # no gloo in this build
        grp, ok = None, 0
    flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device())
                        if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)             # default group: every rank takes part whatever happened above
    if int(flag.item()) == 0:
        grp = None
    _HOST_GROUP["made"], _HOST_GROUP["group"] = True, grp
    return grp


def close_host_group():
    """Destroy the shared host group (call before dist.destroy_process_group())."""
    if _HOST_GROUP["group"] is not None:
        try:
            dist.destroy_process_group(_HOST_GROUP["group"])
        except Exception:
            pass
    _HOST_GROUP["made"], _HOST_GROUP["group"] = False, None


def allreduce_gradients(params, bucket_bytes=64 << 20, comm_dtype=torch.float32):
    """Post-backward form (no overlap): average the gradients of `params` across ranks in flat buckets.  Parameters
    without a gradient on any rank keep grad None; see GradientReducer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    return GradientReducer(params, bucket_bytes=bucket_bytes, comm_dtype=comm_dtype, overlap=False).finish()


def sync_buffers(module, src=0):
    """SyncBatchNorm is deliberately not reproduced (north_star: "gradients only"), so BatchNorm running statistics
    drift per rank; DDP would broadcast rank 0's buffers every iteration (broadcast_buffers=True, train.py:345).  Call
    this before validation and before checkpoint_save so that every rank evaluates -- and the file holds -- ONE model:
    rank `src`'s buffers, in one flat broadcast per dtype."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    for m in module.modules():               # lazily counted num_batches_tracked (MinkowskiEngine/nn.py) -> into the buffers
        if hasattr(m, "flush_ticks"):
            m.flush_ticks()
    groups = {}
    for b in module.buffers():
        groups.setdefault((b.dtype, b.device), []).append(b)
    for (_, _), bufs in sorted(groups.items(), key=lambda kv: str(kv[0])):
        flat = torch.cat([b.detach().reshape(-1) for b in bufs])
        dist.broadcast(flat, src=src)
        off = 0
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    return sum(len(v) for v in groups.values())
