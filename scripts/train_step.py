#!/usr/bin/env python3
"""Training step of BASELINE configs[2]: batch = N ScanNet-sized scenes, one per rank, model_fn forward + losses +
backward, gradient all-reduce over RCCL overlapped with backward (pbnet_amd.dist.GradientReducer; train.py:345), Adam.

    python scripts/train_step.py                       # one rank, process group still initialised through `nccl`
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        scripts/train_step.py --steps 10               # configs[2] on one 8-GPU node

Every rank draws its own synthetic scene (seeds 10 + rank), teacher-forced heads so that the cluster
stage is active from the first step.  Rank 0 prints one JSON line: scenes/s over all ranks (max-over-ranks time), the
all-reduce share, the loss of the last step."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def run_training(world, rank, dev, dist, steps=5, warmup=2, dtype=None, comm_dtype=None, overlap=True, phases_out=None,
                 small=False):
    """configs[2] on this rank's share (one scene), `dist` = an initialised torch.distributed (or None at world 1 without a
    process group: the reducer then has nothing to exchange).  Returns (seconds for `steps` steps on this rank, mean
    all-reduce tail, last loss, scene info).  Importable: bench.py's `train_step` leg calls it."""
    from pbnet_amd import dist as pd, synth
    from pbnet_amd.config import get_config
    from pbnet_amd.network.PBNet import PBNet, model_fn
    dtype = dtype or torch.bfloat16
    comm_dtype = comm_dtype or torch.float32
    cfg = get_config(batch_size=1, cluster_epoch=0)
    torch.manual_seed(22)                                           # same initial weights on every rank
    model = PBNet(cfg).to(dev).train()
    kw = dict(room=(1.6, 1.3, 1.2), n_boxes=4, pitch=0.03, classes=(17, 10)) if small else {}
    batch_np, teacher_np, info = synth.make_train_batch(seed=10 + rank, copies=1, **kw)
    t = torch.from_numpy
    batch = {k: t(v).to(dev) for k, v in batch_np.items()}
    batch["feat_voxel"] = batch["feat_voxel"].to(dtype)
    teacher = {k: t(v).to(dev) for k, v in teacher_np.items()}
    fwd = model.forward
    model.forward = lambda *a, **k: fwd(*a, teacher=teacher, **k)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)   # train.py:331 optim.Adam; one multi-tensor launch per step
    reducer = pd.GradientReducer(model.parameters(), comm_dtype=comm_dtype, overlap=overlap)
    t_comm = [0.0]
    phases = {}

    def mark(name, t_prev, sync):
        if sync:
            torch.cuda.synchronize()
        tt = time.perf_counter()
        phases[name] = phases.get(name, 0.0) + (tt - t_prev)
        return tt

    def step_phases():
        """The same step with a device synchronisation after every phase: where the host waits for the GPU and where the
        GPU waits for the host.  Not the timed configuration."""
        tt = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, pred, visual, meter = model_fn(batch, model, 1, cfg, "train")
        tt = mark("forward_host", tt, False)
        tt = mark("forward_gpu_tail", tt, True)
        loss.backward()
        tt = mark("backward_host", tt, False)
        tt = mark("backward_gpu_tail", tt, True)
        reducer.finish()
        tt = mark("allreduce", tt, True)
        opt.step()
        tt = mark("optimizer_host", tt, False)
        tt = mark("optimizer_gpu_tail", tt, True)
        return loss

    def step():
        # the timed step: no device synchronisation of its own (train.py:383-397 has none either: backward, the reducer's
        # waits -- stream-side for RCCL -- and the optimiser are queued back to back, and the host starts the next forward
        # while the GPU finishes this step).  Round 3 synchronised around finish() to time the all-reduce tail inside the timed
        # loop, which stalled the optimiser's launches and the next forward behind every backward: ~3 ms per step of pure
        # measurement.  The tail is now timed by step_comm() on separate steps.
        opt.zero_grad(set_to_none=True)
        loss, pred, visual, meter = model_fn(batch, model, 1, cfg, "train")
        loss.backward()
        reducer.finish()                                            # waits for the buckets issued during backward
        opt.step()
        return loss

    def step_comm():
        opt.zero_grad(set_to_none=True)
        loss, pred, visual, meter = model_fn(batch, model, 1, cfg, "train")
        loss.backward()
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        reducer.finish()
        torch.cuda.synchronize()
        t_comm[0] += time.perf_counter() - c0
        opt.step()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if dist is not None and dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    comm_steps = min(3, steps)
    for _ in range(comm_steps):                                     # the all-reduce tail (what backward did not hide): untimed steps
        step_comm()
    barrier()
    if phases_out is not None:
        for _ in range(5):
            step_phases()
        phases_out.update({k: v / 5 * 1e3 for k, v in phases.items()})
    # the step's roofline figure: what the three bodies compute in one step (forward, input and weight gradients of every
    # convolution, the batch norms' passes) over the measured step time -- the WHOLE step (coordinates, grouping, glue, losses, Adam
    # included in the time, not in the work): a lower bound of what the kernels reach
    roof = None
    if torch.cuda.is_available() and not small:
        from pbnet_amd.network import train_engine as TE
        TE.ACCOUNTING = []
        step()
        torch.cuda.synchronize()
        recs, TE.ACCOUNTING = TE.ACCOUNTING, None
        if recs:
            acc = TE.step_accounting(recs)
            t_step = elapsed / max(steps, 1)
            roof = {"work": "forward + input gradient + weight gradient of the %d convolutions of the three U-Net bodies (2 x pairs x "
                            "C_in x C_out each) and their batch norms; bytes as SURVEY 8d" % acc["ops"],
                    "flops_per_step": acc["flops_total"], "bytes_per_step": acc["bytes_total"],
                    "flops_by_pass": acc["flops"], "bytes_by_pass": acc["bytes"],
                    "achieved_tflops": round(acc["flops_total"] / t_step / 1e12, 2), "mfma_peak_tflops": 2500.0,
                    "frac_mfma": round(acc["flops_total"] / t_step / 2.5e15, 4),
                    "achieved_GBps": round(acc["bytes_total"] / t_step / 1e9, 1), "hbm_peak_GBps": 8000.0,
                    "frac_hbm": round(acc["bytes_total"] / t_step / 8e12, 4), "bound": "hbm",
                    "note": "over the whole step time (coordinates, grouping, glue, losses and Adam are in the time, not in the work)"}
    if dist is not None and dist.is_initialized():
        pd.sync_buffers(model)                                      # what precedes validation / checkpoint_save
    info = dict(info, reducer=reducer.wire_stats(), roofline=roof)
    return elapsed, t_comm[0] / max(comm_steps, 1), float(loss.detach()), info


def dry_run(args):
    """The N-rank training entry without a GPU: per-rank seeded data, gradient all-reduce through pbnet_amd.dist
    .GradientReducer (hooks during backward, buckets in reverse registration order, a parameter unused on every rank keeps
    grad None), optimizer step, replicas stay bit-identical."""
    import torch.distributed as dist
    from pbnet_amd import dist as pd
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(22)                                           # same initial weights on every rank
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
    unused = torch.nn.Linear(4, 4)                                  # a branch no rank trains (cluster branch before cluster_epoch)
    params = list(model.parameters()) + list(unused.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    reducer = pd.GradientReducer(params, comm_dtype=torch.float32, overlap=not args.no_overlap)
    g = torch.Generator().manual_seed(10 + rank)                    # every rank its own scene
    x, y = torch.randn(64, 16, generator=g), torch.randn(64, 4, generator=g)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        opt.zero_grad(set_to_none=True)
        loss = ((model(x) - y) ** 2).mean()
        loss.backward()
        reducer.finish()
        opt.step()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    lossv = torch.tensor([float(loss)])
    dist.all_reduce(lossv)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "training scenes/s (dry run: stand-in model, launch plumbing only)", "dry_run": True,
                          "n_gpus": world, "steps": args.steps, "value": round(world * args.steps / float(el), 3),
                          "replicas_identical": bool(torch.equal(lo, hi)), "unused_grads_none": all(p.grad is None for p in unused.parameters()),
                          "mean_loss_last_step": round(float(lossv) / world, 6)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--comm-dtype", default="f32", choices=["bf16", "f32"], help="gradient dtype on the wire")
    ap.add_argument("--no-overlap", action="store_true", help="all-reduce after backward instead of from gradient hooks")
    ap.add_argument("--phases", action="store_true", help="also run 5 steps with a synchronisation after every phase and print the split")
    ap.add_argument("--small", action="store_true", help="a 20 k-point room instead of a ScanNet-sized scene (smoke runs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rendezvous / collectives dry run (tests)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: gloo, a small dense stand-in model, the SAME GradientReducer / barriers / MAX-over-ranks / "
                         "JSON line (tests/test_launch_cpu.py)")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dtypes = {"bf16": torch.bfloat16, "f32": torch.float32}
    phases = {} if args.phases else None
    e, comm, loss, info = run_training(world, rank, dev, dist, args.steps, args.warmup, dtypes[args.dtype], dtypes[args.comm_dtype],
                                       not args.no_overlap, phases, args.small)
    el = torch.tensor([e], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if phases and rank == 0:
        print("phases ms/step (synchronised after each): " + ", ".join("%s %.2f" % kv for kv in phases.items()), file=sys.stderr)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    lossv = torch.tensor([loss], device=el.device)
    dist.all_reduce(lossv)
    dist.barrier()
    dist.destroy_process_group()
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)          # RCCL's version banner sits in C stdio: flush it so that the JSON line comes last
    if rank == 0:
        e = float(el.item())
        print(json.dumps({"metric": "training scenes/s (configs[2]: bf16 step, one scene per rank, RCCL gradient all-reduce)",
                          "value": round(world * args.steps / e, 3), "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
                          "ms_per_step": round(e / args.steps * 1e3, 2), "dtype": args.dtype, "comm_dtype": args.comm_dtype,
                          "allreduce_tail_ms_per_step": round(comm * 1e3, 2), "overlap": not args.no_overlap,
                          "mean_loss_last_step": round(float(lossv) / world, 5),
                          "points_per_scene": info["n_points"], "voxels_per_scene": info["n_voxels"],
                          "gradient_buckets": info["reducer"]["buckets"], "bytes_on_the_wire_per_rank_per_step": info["reducer"]["bytes_per_step"],
                          "largest_bucket_bytes": info["reducer"]["largest_bucket_bytes"],
                          "used_flags_over_host_group": info["reducer"]["host_group_for_used_flags"],
                          "roofline": info.get("roofline")}), flush=True)


if __name__ == "__main__":
    main()
