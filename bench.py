#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N MI355X of one node.

metric  : scenes/s of PBNet.forward + binarize + cluster (backbone, heads, grouping, mask U-Net, proposals, score
          U-Net), inputs already voxelised and resident in HBM when the timed region starts.
workload: configs[1] -- one synthetic ScanNet-sized scene (seed 2: 161 517 points, 146 038 voxels at 2 cm),
          MinkUNet34C backbone, bf16 feature slabs with fp32 accumulation, random-init weights (torch.manual_seed(22),
          the reference's init rule).  Randomly initialised heads cannot produce instances, so the semantic / offset
          head outputs are overwritten by teacher-forced values AFTER they have been computed; every
          stage of the path therefore runs inside the timed region on realistic, data-dependent sizes.
step    : one PBNet.forward over one scene per rank.  N > 1: every rank owns its own copy of the scene (identical
          per-GPU work), no data-path collective (scenes are independent at inference) -> weak scaling; the process
          group only carries the barrier and the max-over-ranks of the elapsed time.
in flight: the K steps of a rank are taken round-robin by `--inflight` host threads (default 4), each on its own HIP
          stream: most launches of the path are far too small for 256 CUs, so kernels of independent scenes overlap
          on the device and one scene's host read-backs hide behind another's kernels.  Results are bit-identical to
          the one-at-a-time loop (tests/test_pbnet_gpu.py::test_scenes_in_flight).  `ms_per_step` is elapsed / K (the
          inverse rate); the latency of a scene alone on the GPU is config.one_scene_in_flight_ms_per_scene, and
          `--inflight 1` runs the reference's one-scene-at-a-time loop.

Also on the JSON line:
  roofline     -- the dominant kernel family (k_spconv, csrc/spconv.hip): algorithmic bytes of every launch (SURVEY.md
                  8d: (V_in*C_in + V_out*C_out)*b + K*C_in*C_out*b + 8*P) divided by that launch's duration, measured
                  with HIP events on the launching stream in an instrumented pass over the same steps in the same
                  in-flight mode (a launch that shares the CUs with other streams' kernels takes longer: the per-launch
                  figure falls while the whole-job rate rises; `one_scene_in_flight` holds the same figures for
                  launches that have the GPU to themselves); `traffic` is the HBM bytes per launch from the committed
                  PMC passes (profiles/r01_pmc_summary.json).
  cpu_baseline -- the CPU oracle (oracle/, a restatement: the reference's own CPU path cannot be installed) timed on
                  the host cores of rank 0 at N=1 on one full scene of the same workload.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# one hardware queue per in-flight stream (+ the null stream): the HIP runtime's default of 4 makes a fourth stream share
# a queue with another one (measured: 215 instead of 286 scenes/s); must be set before the runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


WORKLOADS = {
    # C2 = BASELINE configs[1] (the metric's configuration), C4 = configs[3] (dense 1 cm stress scene)
    "c2": dict(seed=2, room=(4.0, 3.2, 2.6), n_boxes=12, pitch=0.0225, voxel=0.02),
    "c4": dict(seed=3, room=(6.4, 5.2, 2.7), n_boxes=14, pitch=0.0112, voxel=0.01),
}


def build_workload(rank, copies, dtype, device, workload="c2"):
    from pbnet_amd import synth
    from pbnet_amd.config import get_config
    from pbnet_amd.network.PBNet import PBNet
    cfg = get_config(test=True)
    torch.manual_seed(22)  # /root/reference/config/config.py:15
    model = PBNet(cfg).to(device).eval()
    # weak scaling: per-GPU work is fixed, so every rank holds its own copy of the SAME scene (other seeds of the
    # generator give scenes of 148-161 k voxels, and the slowest rank would set the time of the whole job)
    w = dict(WORKLOADS[workload])
    batch, teacher, info = synth.make_val_batch(copies=copies, **w)
    b = {k: torch.from_numpy(v).to(device) for k, v in batch.items()}
    b["feat_voxel"] = b["feat_voxel"].to(dtype)
    t = {k: torch.from_numpy(v).to(device) for k, v in teacher.items()}
    return cfg, model, b, t, info, (batch, teacher)


def one_step(model, b, t):
    with torch.no_grad():
        return model(b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"], None, 1, "test", teacher=t)


class ConvProbe(object):
    """Instrumented pass over the same steps: HIP events on the launching stream around every k_spconv launch
    (per-op events inside the native U-Net executor, pbn_unet_forward_timed; torch.cuda.Event around the few launches
    issued from Python) plus each launch's algorithmic bytes / flops."""

    K_OF_KIND = {0: 1, 1: 27, 2: 125, 3: 8, 4: 8}

    def __init__(self):
        self.ms = 0.0
        self.nbytes = 0
        self.flops = 0
        self.launches = 0
        self.py_records = []
        self.lock = threading.Lock()

    def _pairs(self, cm, kind, lin, lout, rows):
        if kind == 0:
            return rows[lout]
        if kind in (3, 4):
            return rows[min(lin, lout)]           # every fine voxel has exactly one parent
        nbr = cm.kernel_map(1 << lout, 3 if kind == 1 else 5)   # cm: the pyramid the executor ran on
        return int((nbr >= 0).sum().item())

    def unet_sink(self, model, plan, rows, cm, esz, op_ms):
        with self.lock:
            self._unet_sink(plan, rows, cm, esz, op_ms)

    def _unet_sink(self, plan, rows, cm, esz, op_ms):
        pair_cache = {}
        widths = [plan["bufs"][i].width for i in range(plan["n_bufs"])]
        for i in range(plan["n_ops"]):
            op = plan["ops"][i]
            key = (op.map_kind, op.level_in, op.level_out)
            if key not in pair_cache:
                pair_cache[key] = self._pairs(cm, op.map_kind, op.level_in, op.level_out, rows)
            pairs = pair_cache[key]
            k = self.K_OF_KIND[op.map_kind]
            cin = op.vpo * (16 // esz)
            cout = op.cout_p
            v_in, v_out = rows[op.level_in], rows[op.level_out]
            nbytes = (v_in * cin + v_out * cout) * esz + k * cin * cout * esz + (8 * pairs if op.map_kind else 0)
            if op.res_buf >= 0:
                nbytes += v_out * cout * esz
            self.nbytes += nbytes
            self.flops += 2 * pairs * cin * cout
            self.ms += op_ms[i]
            self.launches += 1

    def install(self):
        from pbnet_amd.MinkowskiEngine import conv as C
        from pbnet_amd.network import mink_unet as U
        self._orig = C.spconv_forward
        probe = self

        def wrapped(feats, nbr, n_out, packed, **kw):
            w, vpo, n_steps, cout_p = packed
            esz = feats.element_size()
            k = 1 if nbr is None else int(nbr.shape[1])
            cin = int(feats.shape[1])
            pairs = int(n_out) if nbr is None else int((nbr >= 0).sum().item())
            nbytes = (int(feats.shape[0]) * cin + int(n_out) * cout_p) * esz + k * cin * cout_p * esz
            nbytes += 0 if nbr is None else 8 * pairs
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = probe._orig(feats, nbr, n_out, packed, **kw)
            e1.record()
            probe.py_records.append((e0, e1, nbytes, 2 * pairs * cin * cout_p))
            return out

        C.spconv_forward = wrapped
        U.spconv_forward = wrapped
        U.MinkUNet.OP_TIMING_SINK = self.unet_sink
        self._mods = (C, U)

    def remove(self):
        for m in self._mods:
            m.spconv_forward = self._orig
        self._mods[1].MinkUNet.OP_TIMING_SINK = None

    def summary(self):
        torch.cuda.synchronize()
        for e0, e1, nbytes, flops in self.py_records:
            self.ms += e0.elapsed_time(e1)
            self.nbytes += nbytes
            self.flops += flops
            self.launches += 1
        return self.launches, self.ms, self.nbytes, self.flops


def pmc_traffic(args):
    """HBM bytes per k_spconv launch from the committed PMC summary (profiles/r01_pmc_summary.json: two rocprofv3 --pmc
    passes over this same command line, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE).
    Counters cannot be read from inside the process, so the number is only reported for the configuration it was
    collected on (default workload, bf16, 1 copy); anything else -> null."""
    if args.copies != 1 or args.dtype != "bf16" or args.workload != "c2":
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            return int(json.load(f)["k_spconv_traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(cfg, model, raw):
    """Oracle (kind "port") on the host cores: one full scene of the same workload."""
    from oracle import pbnet_ref
    batch, teacher = raw
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    tt = {k: torch.from_numpy(v) for k, v in teacher.items()}
    from pbnet_amd.hostinfo import usable_cores
    cores = usable_cores()
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    s1 = pbnet_ref.backbone_stage(sd, tb["feat_voxel"], tb["xyz_voxel"], tb["v2p_index"])
    s1["sem_pred_score_p"] = tt["sem_score"]
    s1["sem_pred_score_sfp"] = torch.softmax(tt["sem_score"], 1)
    s1["offset_pred_p"] = tt["offset"]
    s1["sem_pred_p"] = tt["sem_score"].max(1)[1]
    pbnet_ref.cluster_stage(sd, cfg, s1, tb["xyz_original"], None, "test")
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": "1 full scene of the same workload (fp32 torch gather-mm-index_add backbone on all host cores + "
                      "single-thread C grouping), %.1f s" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--copies", type=int, default=1, help="rotated copies per scene (reference eval uses 3: TTA)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=4,
                    help="scenes in flight per GPU (one host thread + HIP stream each); 1 = the reference's loop")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = BASELINE configs[1] (default, the metric); c4 = configs[3], the dense 1 cm stress scene")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    cfg, model, b, t, info, raw = build_workload(rank, args.copies, dtype, device, args.workload)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # `--inflight M`: M host threads, each with its own HIP stream, take the K steps round-robin (scenes are independent
    # units, eval_map.py:48-50); kernels of different scenes overlap on the device and one scene's host read-backs hide
    # behind another's kernels.  M = 1 is the reference's one-scene-at-a-time loop.
    streams = [torch.cuda.Stream(device) for _ in range(args.inflight)] if args.inflight > 1 else [None]
    last = [None] * len(streams)

    def run_steps(n):
        if args.inflight == 1:
            for _ in range(n):
                last[0] = one_step(model, b, t)
            return
        errors = []

        def worker(i):
            try:
                torch.cuda.set_device(device)
                with torch.cuda.stream(streams[i]):
                    for _ in range(i, n, args.inflight):
                        last[i] = one_step(model, b, t)
                    streams[i].synchronize()
            except BaseException as e:      # surfaced below: a failed worker must fail the bench
                errors.append(e)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(args.inflight)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if errors:
            raise errors[0]

    one_step(model, b, t)                   # fills the weight / threshold caches once, on one thread
    torch.cuda.synchronize()
    run_steps(args.inflight)                # every stream allocates its scratch (split-K slabs, allocator pools) once
    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    ret = next(r for r in last if r is not None)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n_prop = int(ret["proposals"][1].shape[0] - 1)

    roof = None
    cpu = None
    single_ms = None
    if rank == 0:
        n_probe = max(2, min(args.steps, 5)) * args.inflight

        def probe_leg(runner, n):
            probe = ConvProbe()
            probe.install()
            runner(n)
            n_launch, t_ms, nbytes, flops = probe.summary()
            probe.remove()
            achieved = nbytes / (t_ms * 1e-3) / 1e9
            return {"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "launches_per_step": n_launch // n, "avg_launch_us": round(t_ms * 1e3 / n_launch, 2),
                    "algorithmic_bytes_per_launch": int(nbytes / n_launch),
                    "achieved_tflops": round(flops / (t_ms * 1e-3) / 1e12, 2)}

        # the launches are timed in the mode the timed region ran in: with several scenes in flight a launch shares the
        # CUs with the other streams' kernels, so its own duration grows while the whole-job rate rises
        leg = probe_leg(run_steps, n_probe)
        roof = {"bound": "hbm", "achieved": leg.pop("achieved"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": leg.pop("frac"), "traffic": pmc_traffic(args), "kernel": "k_spconv"}
        roof.update(leg)
        # the same bytes against the wall clock of the whole job (SURVEY.md 8d's path-level form, convolution bytes only):
        # what the GPU sustains across all in-flight scenes, non-convolution stages included in the time
        conv_bytes_per_scene = roof["algorithmic_bytes_per_launch"] * roof["launches_per_step"]
        path_gbs = conv_bytes_per_scene * args.steps / elapsed / 1e9
        roof["path_level"] = {"conv_algorithmic_bytes_per_scene": conv_bytes_per_scene, "achieved": round(path_gbs, 1),
                              "frac": round(path_gbs / HBM_PEAK_GBS, 4)}
        if args.inflight > 1:
            def one_at_a_time(n):
                for _ in range(n):
                    one_step(model, b, t)
            roof["one_scene_in_flight"] = probe_leg(one_at_a_time, max(2, min(args.steps, 5)))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            one_at_a_time(10)
            torch.cuda.synchronize()
            single_ms = (time.perf_counter() - t1) / 10 * 1e3
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, model, raw)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line = {
            "metric": "scenes/sec fwd+cluster (ScanNet ~150k pts/scene)",
            "value": round(world * args.steps / elapsed, 3),
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic room scene (pbnet_amd/synth.py), random-init weights (seed 22), teacher-forced "
                    "semantic/offset head outputs",
            "config": {"workload": "%s: 1 scene, %d pts, %d voxels @%gcm, %d rotated cop%s, full "
                                   "PBNet.forward (MinkUNet34C + grouping + MinkUNet14A mask + MinkUNet34C score)"
                                   % ("configs[1]" if args.workload == "c2" else "configs[3]",
                                      info["n_points"] // args.copies, info["n_voxels"] // args.copies,
                                      WORKLOADS[args.workload]["voxel"] * 100, args.copies,
                                      "y" if args.copies == 1 else "ies"),
                       "points_per_step": info["n_points"], "voxels_per_step": info["n_voxels"],
                       "proposals_per_step": n_prop, "scenes_in_flight_per_gpu": args.inflight,
                       "one_scene_in_flight_ms_per_scene": None if single_ms is None else round(single_ms, 3),
                       "parallelism": "scenes sharded over GPUs, %d in flight per GPU (host thread + HIP stream each), "
                                      "no data-path collective" % args.inflight},
            "roofline": roof,
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))


if __name__ == "__main__":
    main()
