#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N MI355X of one node.

metric  : scenes/s of PBNet.forward + binarize + cluster (backbone, heads, grouping, mask U-Net, proposals, score
          U-Net), inputs already voxelised and resident in HBM when the timed region starts.
workload: configs[1] -- one synthetic ScanNet-sized scene (seed 2: 161 517 points, 146 038 voxels at 2 cm),
          MinkUNet34C backbone, bf16 feature slabs with fp32 accumulation, random-init weights (torch.manual_seed(22),
          the reference's init rule).  Randomly initialised heads cannot produce instances, so the semantic / offset
          head outputs are overwritten by teacher-forced values AFTER they have been computed; every
          stage of the path therefore runs inside the timed region on realistic, data-dependent sizes.
step    : one PBNet.forward over one scene per rank (1 rotated copy).  N > 1: every rank owns its own copy of the scene
          (identical per-GPU work), no data-path collective (scenes are independent at inference) -> weak scaling; the
          process group (RCCL) only carries the barrier and the max-over-ranks of the elapsed time.  The process group is
          initialised at N = 1 too (one all-reduce + barrier) so that the RCCL path is known to work on the box.
timing  : W warm-up steps, then `--repeats` timed blocks of EXACTLY K steps, each bracketed by barrier +
          torch.cuda.synchronize() on both sides and reduced with MAX over ranks.  `value` / `ms_per_step` come from the
          MEDIAN block; p10 / p90 over the blocks are reported beside it.
in flight: the K steps of a rank are taken round-robin by `--inflight` host threads (default 4), each on its own HIP
          stream: most launches of the path are far too small for 256 CUs, so kernels of independent scenes overlap
          on the device and one scene's host read-backs hide behind another's kernels.  Results are bit-identical to
          the one-at-a-time loop (tests/test_pbnet_gpu.py::test_scenes_in_flight).  `ms_per_step` is elapsed / K (the
          inverse rate); the latency of a scene alone on the GPU is config.one_scene_in_flight_ms_per_scene, and
          `--inflight 1` runs the reference's one-scene-at-a-time loop.

Also on the JSON line:
  roofline     -- the dominant kernel family (k_spconv*, csrc/spconv.hip): algorithmic bytes of every launch
                  ((V_in*C_in + V_out*C_out)*b + K*C_in*C_out*b + 8*P, true channel counts, no residual term) divided by
                  that launch's duration, measured with HIP events on the launching stream in an instrumented pass over
                  the same steps in the same in-flight mode; `by_level` splits the figure per tensor stride (the
                  stride-1/2 levels carry 83 % of the activation bytes); `one_scene_in_flight` holds the same figures for
                  launches that have the GPU to themselves; `traffic` is the HBM bytes per launch from the committed PMC
                  passes (profiles/*_pmc_summary.json).
  stages_ms    -- per-stage wall time of one scene alone on the GPU (device synchronised at the stage boundaries).
  grouping     -- points grouped per forward, microseconds and points/s of the grouping stage.
  tta3         -- the same path on the reference's 3-rotated-copies eval batch (dataset_preprocess.py:324).
  cpu_baseline -- the CPU oracle (oracle/, a restatement: the reference's own CPU path cannot be installed) timed on
                  the host cores of rank 0 at N=1 on full scenes of the same workload, median of 3.
  phases_s     -- wall clock of the phases of this run (build, warm-up, timed blocks, probes, TTA leg, CPU baseline).
"""
import argparse
import contextlib
import json
import os
import sys
import threading
import time

T_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# one hardware queue per in-flight stream (+ the null stream): the HIP runtime's default of 4 makes a fourth stream share
# a queue with another one (measured: 215 instead of 286 scenes/s); must be set before the runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
PMC_SUMMARIES = ("r06_b_pmc_summary.json", "r06_a_pmc_summary.json", "r05_f_pmc_summary.json", "r05_e_pmc_summary.json", "r05_d_pmc_summary.json", "r05_c_pmc_summary.json", "r05_b_pmc_summary.json", "r05_a_pmc_summary.json", "r04_d_pmc_summary.json", "r04_c_pmc_summary.json", "r04_b_pmc_summary.json", "r04_a_pmc_summary.json", "r03_c_pmc_summary.json", "r03_b_pmc_summary.json", "r03_a_pmc_summary.json", "r02_b_pmc_summary.json", "r02_a_pmc_summary.json", "r01_pmc_summary.json")   # newest first


WORKLOADS = {
    # C2 = BASELINE configs[1] (the metric's configuration), C4 = configs[3] (dense 1 cm stress scene)
    "c2": dict(seed=2, room=(4.0, 3.2, 2.6), n_boxes=12, pitch=0.0225, voxel=0.02),
    "c4": dict(seed=3, room=(6.4, 5.2, 2.7), n_boxes=14, pitch=0.0112, voxel=0.01),
}


def build_model(device):
    from pbnet_amd.config import get_config
    from pbnet_amd.network.PBNet import PBNet
    cfg = get_config(test=True)
    torch.manual_seed(22)  # /root/reference/config/config.py:15
    return cfg, PBNet(cfg).to(device).eval()


def pin_rank_to_cpus(local_rank, ranks_on_node):
    """Before any GPU call: confine this rank (its 4 scene threads, the HIP runtime's helper threads) to its own contiguous
    share of the host cores -- contiguous core ranges follow the NUMA nodes on the two-socket MI355X hosts, and 8 ranks x 4
    threads must not migrate across sockets.  PBN_BENCH_PIN=0 switches it off; returns the core range or None."""
    if os.environ.get("PBN_BENCH_PIN", "1") == "0" or ranks_on_node <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // ranks_on_node
        if per < 2:                         # a rank's four scene threads on one core would be worse than no pinning
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return [mine[0], mine[-1]]
    except OSError:
        return None


def build_scene(copies, dtype, device, workload="c2", seed=None):
    from pbnet_amd import synth
    # one rank: the scene of BASELINE configs[1] (seed 2).  N ranks: rank r takes seed 10 + r (seeds 10-17) --
    # distinct scenes of 148-161 k voxels, so the job's time is the slowest rank's (MAX over ranks) and per-rank rates are
    # reported next to it
    kw = dict(WORKLOADS[workload])
    if seed is not None:
        kw["seed"] = seed
    batch, teacher, info = synth.make_val_batch(copies=copies, **kw)
    b = {k: torch.from_numpy(v).to(device) for k, v in batch.items()}
    b["feat_voxel"] = b["feat_voxel"].to(dtype)
    # the teacher-forced head outputs stand in for slabs the path itself would hold in `dtype`: resident in that dtype
    t = {k: torch.from_numpy(v).to(device).to(dtype) for k, v in teacher.items()}
    return b, t, info, (batch, teacher)


def build_batched(seeds, dtype, device):
    """Several DISTINCT scenes in one forward through the batch index (the reference's own batch axis: PBNet.py:167-170,
    dataset_preprocess.py:296): scene j of `seeds` becomes batch element j."""
    from pbnet_amd import synth
    parts = [synth.make_val_batch(copies=1, **dict(WORKLOADS["c2"], seed=sd)) for sd in seeds]
    vox, feat, xyz, v2p, off, score = [], [], [], [], [], []
    nv = 0
    for j, (bt, tc, _) in enumerate(parts):
        xv = bt["xyz_voxel"].copy()
        xv[:, 0] = j
        vox.append(xv); feat.append(bt["feat_voxel"]); xyz.append(bt["xyz_original"]); v2p.append(bt["v2p_index"] + nv)
        off.append(tc["offset"]); score.append(tc["sem_score"])
        nv += len(xv)
    b = {"xyz_voxel": torch.from_numpy(np.concatenate(vox)).to(device),
         "feat_voxel": torch.from_numpy(np.concatenate(feat)).to(device).to(dtype),
         "xyz_original": torch.from_numpy(np.concatenate(xyz)).to(device),
         "v2p_index": torch.from_numpy(np.concatenate(v2p)).to(device)}
    t = {"sem_score": torch.from_numpy(np.concatenate(score)).to(device), "offset": torch.from_numpy(np.concatenate(off)).to(device)}
    info = {"n_points": int(sum(p[2]["n_points"] for p in parts)), "n_voxels": int(nv), "scenes": len(seeds)}
    return b, t, info


def timed_leg(model, b, t, inflight, device, steps, warmup=None, blocks=3):
    """scenes-forwards per second of a side configuration (median of `blocks` timed blocks) + one forward alone, in ms."""
    r = Runner(model, b, t, inflight, device)
    r.run(inflight)
    r.run(max(2, steps // 4) if warmup is None else warmup)
    bl = []
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(steps)
        torch.cuda.synchronize()
        bl.append(time.perf_counter() - t0)
    e = float(np.median(bl))
    for _ in range(2):                         # this thread's stream has its own allocator pool: fill it outside the timing
        one_step(model, b, t)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        one_step(model, b, t)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t1) / 5 * 1e3
    return steps / e, e / steps * 1e3, alone, r.result()


def served_legs(model, dtype, device, inflight, seconds=1.2):
    """scenes/s of the serving front over a grid of (scenes per forward B, forwards in flight F); the best cell is `value`."""
    from pbnet_amd import synth
    from pbnet_amd.serving import SceneServer
    scenes = []
    for sd in range(2, 10):
        bt, tc, inf = synth.make_val_batch(copies=1, **dict(WORKLOADS["c2"], seed=sd))
        sc = {k: torch.from_numpy(bt[k]).to(device) for k in ("xyz_voxel", "feat_voxel", "xyz_original", "v2p_index")}
        sc["feat_voxel"] = sc["feat_voxel"].to(dtype)
        scenes.append((sc, {k: torch.from_numpy(v).to(device) for k, v in tc.items()}, inf["n_points"]))
    torch.cuda.synchronize()
    streams = inflight_streams(device, inflight)
    grid, best = {}, None
    for B, F in ((1, 4), (2, 2), (2, 4), (3, 3), (3, 4), (4, 1), (4, 2), (4, 3), (4, 4), (6, 2), (6, 3), (8, 1), (8, 2)):
        if F > len(streams):
            continue
        srv = SceneServer(model, max_batch=B, forwards_in_flight=F, streams=streams[:F])
        try:
            def burst(n):
                futs = [srv.submit(scenes[i % 8][0], scenes[i % 8][1]) for i in range(n)]
                return [f.result(timeout=600) for f in futs]
            burst(2 * B * F)                                  # warm-up: pools grow to the merged sizes
            n = 2 * B * F
            torch.cuda.synchronize()
            t0 = time.perf_counter(); burst(n); dt = time.perf_counter() - t0
            n = max(2 * B * F, int(n * seconds / max(dt, 1e-3)) // (B * F) * (B * F))
            f0 = srv.forwards
            torch.cuda.synchronize()
            t0 = time.perf_counter(); res = burst(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            cell = {"value": round(n / dt, 3), "scenes": n, "forwards": srv.forwards - f0,
                    "proposals_per_scene": round(float(np.mean([int(r["proposals"][1].shape[0]) - 1 for r in res])), 1)}
        finally:
            srv.close()
        grid["B%dxF%d" % (B, F)] = cell
        if best is None or cell["value"] > best[1]["value"]:
            best = ("B%dxF%d" % (B, F), cell)
        torch.cuda.empty_cache()
    return {"value": best[1]["value"], "unit": "scenes/s", "best": best[0], "grid": grid,
            "scene_stream": "8 distinct configs[1] scenes (seeds 2..9, %d-%d points), submitted in bursts; every scene gets its own result back "
                            "(proposals renumbered, point indices local)" % (min(s[2] for s in scenes), max(s[2] for s in scenes)),
            "note": "pbnet_amd/serving.py: B = scenes merged per forward through the reference's batch axis (PBNet.py:167-176, "
                    "dataset_preprocess.py:296), F = forwards in flight; B1xF4 is the headline's mode with distinct scenes and the split"}


def build_workload(rank, copies, dtype, device, workload="c2", world=1):
    cfg, model = build_model(device)
    b, t, info, raw = build_scene(copies, dtype, device, workload, seed=(10 + rank) if world > 1 else None)
    return cfg, model, b, t, info, raw


_STREAMS = {}


def inflight_streams(device, n):
    """The `n` HIP streams every in-flight leg of this process runs on, created once.  Round 5: streams created later in the
    process can land on a hardware queue that another stream of the same leg already uses (the runtime hands its 8 queues
    out as streams are first used, then doubles up) -- two scenes of a leg then run one after the other and the leg measures
    ~20 % low (283 instead of 350 scenes/s on the second set of four streams of a process, reproducibly)."""
    key = (str(device), n)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device) for _ in range(n)]
        for st in _STREAMS[key]:        # first USE is what binds a stream to a hardware queue: do it now, in this order, before
            with torch.cuda.stream(st):  # anything else of the process (graph capture side streams, ...) takes queues in between
                torch.zeros(16, device=device).add_(1)
        torch.cuda.synchronize(device)
    return _STREAMS[key]


def planned_leg(model, b, t, dtype, inflight, device, steps, graph):
    """The sync-free forward (pbnet_amd/planned.py: every data-dependent size stays on the device) on `inflight` streams, eager
    launch sequence or HIP-graph replay: scenes/s, and one scene alone in ms.  The difference to the headline (the size-exact
    forward with its five read-backs: three pyramids' level counts, the front, the proposal counts) is what the host costs."""
    r = Runner(model, b, t, inflight, device, mode="graph" if graph else "planned", dtype=dtype)
    r.run(2 * inflight)
    bl = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(steps)
        torch.cuda.synchronize()
        bl.append(time.perf_counter() - t0)
    e = float(np.median(bl))
    with torch.cuda.stream(r.streams[0]) if r.streams[0] is not None else contextlib.nullcontext():
        r.run(3, threads=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r.run(8, threads=1)
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t1) / 8 * 1e3
        out = r.step(0)
        torch.cuda.synchronize()
    finite = bool(torch.isfinite(out["clt_scores"].float()).all()) if out["clt_scores"].numel() else True
    return {"value": round(steps / e, 3), "unit": "scenes/s", "ms_per_step": round(e / steps * 1e3, 3),
            "scenes_in_flight": inflight, "one_scene_in_flight_ms_per_scene": round(alone, 3),
            "proposals_per_step": int(out["proposals"][1].shape[0] - 1), "finite": finite}


def one_step(model, b, t):
    with torch.no_grad():
        return model(b["feat_voxel"], b["xyz_voxel"], b["xyz_original"], b["v2p_index"], None, 1, "test", teacher=t)


class ConvProbe(object):
    """Instrumented pass over the same steps: HIP events on the launching stream around every convolution op (per-op
    events inside the native U-Net executor, pbn_unet_forward_timed; torch.cuda.Event around the few launches issued
    from Python) plus each op's algorithmic bytes / flops."""

    K_OF_KIND = {0: 1, 1: 27, 2: 125, 3: 8, 4: 8}

    def __init__(self):
        self.ms = 0.0
        self.nbytes = 0          # strict SURVEY 8d bytes: true channel counts, no residual term
        self.nbytes_r1 = 0       # round-1 accounting (8-padded C_in, + residual read): kept for comparison
        self.flops = 0
        self.launches = 0
        self.levels = {}         # tensor stride of the output level -> [launches, ms, bytes, flops]
        self.families = {}       # kernel family (pbn_spconv_family) -> [launches, ms, bytes, flops]
        self.py_records = []
        self.lock = threading.Lock()

    def _pairs(self, cm, kind, lin, lout, rows):
        if kind == 0:
            return rows[lout]
        if kind in (3, 4):
            return rows[min(lin, lout)]           # every fine voxel has exactly one parent
        nbr = cm.kernel_map(1 << lout, 3 if kind == 1 else 5)   # cm: the pyramid the executor ran on
        return int((nbr >= 0).sum().item())

    FAMILY_NAMES = {0: "workgroup-tile (k_spconv)", 1: "wave-autonomous (k_spconv_wave)", 2: "row-stationary (k_spconv_rs)"}

    def _family(self, n_out, k, vpo, n_steps, cout_p, esz, has_map):
        from pbnet_amd import _native as N
        dt = {4: 0, 2: 1}[esz]            # PBN_F32 / PBN_BF16 (fp16 chooses as bf16)
        return self.FAMILY_NAMES.get(int(N.lib().pbn_spconv_family(int(n_out), int(k), int(vpo), int(n_steps), int(cout_p), dt, int(bool(has_map)))), "?")

    def unet_sink(self, model, plan, rows, cm, esz, op_ms):
        with self.lock:
            self._unet_sink(plan, rows, cm, esz, op_ms)

    def _unet_sink(self, plan, rows, cm, esz, op_ms):
        pair_cache = {}
        second = {i: c for i, c, _ in plan.get("folded_io", [])}     # op index -> C_in of a folded 1x1 shortcut (second source)
        for i in range(plan["n_ops"]):
            op = plan["ops"][i]
            key = (op.map_kind, op.level_in, op.level_out)
            if key not in pair_cache:
                pair_cache[key] = self._pairs(cm, op.map_kind, op.level_in, op.level_out, rows)
            pairs = pair_cache[key]
            k = self.K_OF_KIND[op.map_kind]
            cin_t, cout_t = plan["true_io"][i]
            cin_p, cout_p = op.vpo * (16 // esz), op.cout_p
            v_in, v_out = rows[op.level_in], rows[op.level_out]
            nbytes = (v_in * cin_t + v_out * cout_t) * esz + k * cin_t * cout_t * esz + (8 * pairs if op.map_kind else 0)
            r1 = (v_in * cin_p + v_out * cout_p) * esz + k * cin_p * cout_p * esz + (8 * pairs if op.map_kind else 0)
            if op.res_buf >= 0:
                r1 += v_out * cout_p * esz
            flops = 2 * pairs * cin_t * cout_t
            if i in second:              # the folded shortcut: its input slab and its weights are read as well, nothing more is written
                extra = (v_out * second[i] + second[i] * cout_t) * esz
                nbytes += extra
                r1 += extra
                flops += 2 * v_out * second[i] * cout_t
            self.nbytes += nbytes
            self.nbytes_r1 += r1
            self.flops += flops
            self.ms += op_ms[i]
            self.launches += 1
            lv = self.levels.setdefault(1 << op.level_out, [0, 0.0, 0, 0])
            lv[0] += 1; lv[1] += op_ms[i]; lv[2] += nbytes; lv[3] += flops
            fam = self._family(v_out, k, op.vpo, op.n_steps, op.cout_p, esz, op.map_kind != 0)
            fv = self.families.setdefault(fam, [0, 0.0, 0, 0])
            fv[0] += 1; fv[1] += op_ms[i]; fv[2] += nbytes; fv[3] += flops

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
            self.nbytes_r1 += nbytes
            self.flops += flops
            self.launches += 1
        return self.launches, self.ms, self.nbytes, self.flops


def pmc_traffic(args):
    """HBM bytes per convolution launch from the newest committed PMC summary (two rocprofv3 --pmc passes over this
    same command line, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE).  Counters cannot
    be read from inside the process, so the number is only reported for the configuration it was collected on (default
    workload, bf16, 1 copy); anything else -> null."""
    if args.copies != 1 or args.dtype != "bf16" or args.workload != "c2":
        return None, None
    for name in PMC_SUMMARIES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return int(json.load(f)["k_spconv_traffic_bytes_per_launch"]), "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def gpu_active(args):
    """gpu_active_frac: union of kernel intervals / wall from the newest committed kernel-trace summary of
    this command line (scripts/profile_round.sh -> profiles/<tag>_bench_concurrency.json for the in-flight mode,
    <tag>_inflight1_concurrency.json for one scene alone, <tag>_planned_concurrency.json / <tag>_graph_concurrency.json for
    `--forward-mode planned / graph` with four scenes in flight).  A trace cannot be taken from inside the process."""
    if args.copies != 1 or args.dtype != "bf16" or args.workload != "c2":
        return None
    out = {}
    for tag in ("r06_b", "r06_a", "r05_f", "r05_e", "r05_d", "r05_c", "r05_b", "r05_a"):
        for key, name in (("in_flight", "%s_bench_concurrency.json" % tag), ("one_scene", "%s_inflight1_concurrency.json" % tag),
                          ("in_flight_planned_eager", "%s_planned_concurrency.json" % tag),
                          ("in_flight_graph", "%s_graph_concurrency.json" % tag)):
            if key in out:
                continue
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    d = json.load(f)
                out[key] = {"gpu_active_frac": d["gpu_active_frac"], "mean_kernels_running": d["mean_kernels_running"],
                            "source": "profiles/" + name}
            except (OSError, KeyError, ValueError):
                continue
    return out or None


def sq_summary():
    """(groups, source) of the newest committed SQ counter summary that carries MFMA busy cycles per launch (round 6 on), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_conv_summary.json")), reverse=True):
        try:
            with open(path) as f:
                g = json.load(f)["groups"]
            if any("mfma_busy_cycles_per_launch" in v for v in g.values()):
                return g, "profiles/" + os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline(cfg, model, raw, runs=3):
    """Oracle (kind "port") on the host cores: full scenes of the same workload, median of `runs`."""
    from oracle import pbnet_ref
    batch, teacher = raw
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    tt = {k: torch.from_numpy(v) for k, v in teacher.items()}
    from pbnet_amd.hostinfo import usable_cores
    cores = usable_cores()
    torch.set_num_threads(cores)
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        s1 = pbnet_ref.backbone_stage(sd, tb["feat_voxel"], tb["xyz_voxel"], tb["v2p_index"])
        s1["sem_pred_score_p"] = tt["sem_score"]
        s1["sem_pred_score_sfp"] = torch.softmax(tt["sem_score"], 1)
        s1["offset_pred_p"] = tt["offset"]
        s1["sem_pred_p"] = tt["sem_score"].max(1)[1]
        pbnet_ref.cluster_stage(sd, cfg, s1, tb["xyz_original"], None, "test")
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return {"value": 1.0 / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": "%d full scenes of the same workload, median (fp32 torch gather-mm-index_add backbone on all host "
                      "cores + single-thread C grouping): %s s" % (runs, ", ".join("%.1f" % t for t in times))}


def grouped_points(raw):
    """Points that enter the grouping stage: the classes that pass the population gate (PBNet.py:156)."""
    from pbnet_amd.network.PBNet import COUNT_MEAN
    sem = raw[1]["sem_score"].argmax(1)
    cnt = np.bincount(sem, minlength=20)
    return int(sum(int(cnt[c]) for c in range(2, 20) if not (float(cnt[c]) < np.float32(COUNT_MEAN[c]) * np.float32(0.05))))


class Runner(object):
    """K steps taken round-robin by `inflight` host threads, one HIP stream each (inflight 1: a plain loop).
    mode "size-exact": `model(...)` (the headline); "planned" / "graph": the sync-free forward of pbnet_amd/planned.py on
    capacities of 1.25 x this scene's measured sizes, issued as eager launches / replayed from one HIP graph per stream."""

    def __init__(self, model, b, t, inflight, device, mode="size-exact", dtype=None):
        self.model, self.b, self.t, self.inflight, self.device, self.mode = model, b, t, inflight, device, mode
        self.streams = inflight_streams(device, inflight) if inflight > 1 else [None]
        self.last = [None] * len(self.streams)
        self.pfs = None
        if mode != "size-exact":
            from pbnet_amd import planned
            dtype = dtype or b["feat_voxel"].dtype
            self.args = (b["feat_voxel"].to(dtype), b["xyz_voxel"], b["xyz_original"], b["v2p_index"])
            cap = planned.measure_capacities(model, *self.args, teacher=t).padded(1.25)
            self.pfs = []
            for st in self.streams:
                with torch.cuda.stream(st if st is not None else torch.cuda.current_stream(device)):
                    pf = planned.PlannedForward(model, cap, dtype=dtype)
                    pf(*self.args, teacher=t)
                    if mode == "graph":
                        pf.capture(*self.args, teacher=t)
                    self.pfs.append(pf)
                torch.cuda.synchronize()

    def step(self, i):
        if self.pfs is None:
            return one_step(self.model, self.b, self.t)
        pf = self.pfs[i]
        return pf.finish(pf.replay()) if self.mode == "graph" else pf(*self.args, teacher=self.t)

    def run(self, n, threads=None):
        m = self.inflight if threads is None else threads
        if m <= 1:
            for _ in range(n):
                self.last[0] = self.step(0)
            if self.streams[0] is not None:
                torch.cuda.synchronize()
            return
        errors = []

        def worker(i):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.streams[i]):
                    for _ in range(i, n, m):
                        self.last[i] = self.step(i)
                    self.streams[i].synchronize()
            except BaseException as e:      # surfaced below: a failed worker must fail the bench
                errors.append(e)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(m)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if errors:
            raise errors[0]

    def result(self):
        return next(r for r in self.last if r is not None)


def init_process_group(world, rank, device):
    """RCCL process group (backend "nccl" is RCCL on ROCm), also at world size 1: init + one all-reduce + barrier, so the
    multi-GPU code path has run on this box before an 8-GPU node ever sees it.  At N = 1 a failure is reported on the JSON
    line instead of failing the run (the single-GPU metric does not need the group)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1 and "MASTER_PORT" not in os.environ:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        probe = torch.full((1024,), float(rank + 1), device=device)
        dist.all_reduce(probe)
        dist.barrier()
        torch.cuda.synchronize()
        want = world * (world + 1) / 2
        if float(probe[0]) != want:
            raise RuntimeError("all-reduce returned %r, expected %r" % (float(probe[0]), want))
        check = rccl_selfcheck(dist, world, rank, device)
        return dist, {"status": "ok (backend nccl = RCCL, world %d: init, all-reduce, barrier)" % world, "selfcheck": check}
    except Exception as e:  # noqa: BLE001
        if world > 1:
            raise
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        return None, "FAILED at world 1: %s: %s" % (type(e).__name__, str(e)[:200])


def rccl_selfcheck(dist, world, rank, device, mb=64):
    """Round 6: what the first real N-GPU run must show in its own output -- that RCCL moved a gradient-bucket-sized buffer
    between N ranks and every rank got the right sum.  Element i of rank r's 64 MB fp32 bucket is (r + 1) * (1 + i % 7); after the
    all-reduce every element must be W (W + 1) / 2 * (1 + i % 7) (small integers: exact in fp32), checked on EVERY rank; the
    per-rank verdicts and times are all-gathered so that rank 0's JSON line carries them, and every rank prints its own line
    to stderr.  Bus bandwidth = 2 (W - 1) / W x bytes / time (the ring all-reduce's traffic per link)."""
    n = mb * (1 << 20) // 4
    device = torch.device(device)
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else (lambda: None)
    pattern = (torch.arange(n, device=device, dtype=torch.int32) % 7 + 1).float()
    times = []
    ok = True
    for _ in range(3):
        buf = pattern * float(rank + 1)
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        dist.all_reduce(buf)
        sync()
        times.append(time.perf_counter() - t0)
        ok = ok and bool(torch.equal(buf, pattern * float(world * (world + 1) // 2)))
    mine = torch.tensor([float(rank), 1.0 if ok else 0.0, min(times) * 1e3], device=device)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    rows = [[float(v) for v in t.cpu().tolist()] for t in allr]
    sys.stderr.write("rccl self-check rank %d/%d: %s, %d MB all-reduce %.3f ms\n" % (rank, world, "ok" if ok else "WRONG SUM", mb, min(times) * 1e3))
    sys.stderr.flush()
    t_max = max(r[2] for r in rows) / 1e3
    return {"bucket_mb": mb, "ranks_seen": len({int(r[0]) for r in rows}), "all_ok": all(r[1] == 1.0 for r in rows),
            "per_rank": [{"rank": int(r[0]), "ok": r[1] == 1.0, "ms": round(r[2], 3)} for r in rows],
            "bus_gb_per_s": round(2.0 * (world - 1) / world * mb * (1 << 20) / max(t_max, 1e-9) / 1e9, 1) if world > 1 else None}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- plain child processes with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, exactly what `python -m torch.distributed.run --nproc-per-node N` would give them --
    from a process that has not touched the GPU (it never will: it only waits), forward rank 0's JSON line, and exit with the
    worst child's code.  (Never exec from a process that has initialised the GPU; never re-exec: children are children.)"""
    import socket
    import subprocess
    n = args.gpus
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [q.wait() for q in procs[1:]]
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    worst = max((abs(c) for c in codes), default=0)
    if worst:
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
    sys.exit(0 if worst == 0 else 1)


def dry_run(args):
    """The launch contract without a GPU: WORLD_SIZE / RANK / LOCAL_RANK / MASTER_* from the environment, one process per
    rank, warmup + EXACTLY K steps between barriers, MAX over ranks, ONE JSON line from rank 0.  The step is a stand-in (a
    per-rank seeded numpy reduction); every field that describes the device path is marked as not measured."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pbnet_amd.dist import shard_scenes
    mine = shard_scenes(8 * world, rank, world)                         # the scene shard this rank would take (eval_map.py:48-50)
    rng = np.random.default_rng(1000 + rank)
    work = rng.standard_normal(200000)

    def step():
        return float(np.sort(work)[::97].sum())

    def run(n):           # n steps over --inflight host threads, as the real runner issues them (one thread per scene in flight)
        if args.inflight <= 1:
            for _ in range(n):
                step()
            return
        counts = [n // args.inflight + (1 if i < n % args.inflight else 0) for i in range(args.inflight)]
        th = [threading.Thread(target=lambda c=c: [step() for _ in range(c)]) for c in counts]
        for x in th:
            x.start()
        for x in th:
            x.join()

    run(args.warmup)
    dist.barrier()
    t0 = time.perf_counter()
    run(args.steps)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    shard_sizes = torch.zeros(world, dtype=torch.int64)
    shard_sizes[rank] = len(mine)
    dist.all_reduce(shard_sizes)
    check = rccl_selfcheck(dist, world, rank, "cpu", mb=4)      # the same self-check the GPU run makes over RCCL, here over gloo
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        e = float(el.item())
        print(json.dumps({"metric": "scenes/sec fwd+cluster (ScanNet ~150k pts/scene)", "value": round(world * args.steps / e, 3),
                          "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(e / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": args.dtype, "data": "none (dry run)", "dry_run": True,
                          "config": {"workload": "stand-in step on the host: launch plumbing only, NOT a measurement",
                                     "local_rank_of_rank0": local_rank, "scene_shard_sizes": shard_sizes.tolist(),
                                     "host_threads_per_rank": max(1, args.inflight),
                                     "rccl": {"status": "not used (gloo)", "selfcheck": check}},
                          "roofline": None}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps at least; the median block is reported")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="keep adding timed blocks (EXACTLY --steps steps each) until this much timed wall has accumulated "
                         "(VERDICT round 4: 5 blocks of 0.06 s carried 5-10 %% of noise); at most --max-blocks blocks")
    ap.add_argument("--max-blocks", type=int, default=64)
    ap.add_argument("--copies", type=int, default=1, help="rotated copies per scene (reference eval uses 3: TTA)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stage breakdown, the TTA leg and the CPU baseline")
    ap.add_argument("--switch-interval", type=float, default=float(os.environ.get("PBN_BENCH_SWITCH", "0") or 0),
                    help="sys.setswitchinterval for the in-flight host threads (0 = Python's default 5 ms)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="scenes in flight per GPU (one host thread + HIP stream each); 1 = the reference's loop")
    ap.add_argument("--forward-mode", default="size-exact", choices=["size-exact", "planned", "graph"],
                    help="what a step runs: model(...) (the headline, default); the sync-free forward of pbnet_amd/planned.py as "
                         "eager launches; the same replayed from one HIP graph per stream (for profiles of those modes: use with --no-extras)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = BASELINE configs[1] (default, the metric); c4 = configs[3], the dense 1 cm stress scene")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: rendezvous over gloo, a stand-in step, the same barriers / MAX-over-ranks timing / JSON line "
                         "(tests/test_launch_cpu.py proves the N-rank plumbing before an 8-GPU node runs it)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)          # `python bench.py --gpus N` by itself: one child per rank, started BEFORE any GPU call
    if args.dry_run:
        return dry_run(args)
    phases = {}
    t_phase = [time.perf_counter()]

    def phase(name):
        now = time.perf_counter()
        phases[name] = round(phases.get(name, 0.0) + now - t_phase[0], 2)
        t_phase[0] = now

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pinned = pin_rank_to_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))   # before anything touches the GPU
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    assert world == args.gpus, "WORLD_SIZE %d does not match --gpus %d (launch with torch.distributed.run --nproc-per-node %d, or run `python bench.py --gpus %d` by itself: it starts its ranks)" % (world, args.gpus, args.gpus, args.gpus)
    phases["import_torch"] = round(time.perf_counter() - T_START, 2)
    t_phase[0] = time.perf_counter()
    dist, rccl_status = init_process_group(world, rank, device)
    phase("rccl_init")

    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    cfg, model, b, t, info, raw = build_workload(rank, args.copies, dtype, device, args.workload, world)
    # the headline is the SIZE-EXACT forward (every scene new to the model, as in the reference's test loop)
    phase("build_workload")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.switch_interval > 0:
        sys.setswitchinterval(args.switch_interval)
    one_step(model, b, t)                   # fills the weight / threshold caches once, on one thread
    torch.cuda.synchronize()
    runner = Runner(model, b, t, args.inflight, device, mode=args.forward_mode)
    runner.run(args.inflight)               # every stream allocates its scratch and allocator pools once
    phase("first_steps")
    runner.run(args.warmup)
    blocks, own_blocks = [], []
    while True:
        if len(blocks) >= max(1, args.repeats):
            # every rank must take the same decision: rank 0's clock decides (the blocks' MAX-over-ranks times are equal on all ranks)
            enough = sum(blocks) >= args.min_seconds or len(blocks) >= args.max_blocks
            if dist is not None:
                flag = torch.tensor([1 if enough else 0], dtype=torch.int32, device=device)
                dist.broadcast(flag, src=0)
                enough = bool(flag.item())
            if enough:
                break
        barrier()
        t0 = time.perf_counter()
        runner.run(args.steps)              # EXACTLY K steps
        torch.cuda.synchronize()
        own_blocks.append(time.perf_counter() - t0)      # this rank's own steps, before it waits for the others
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        blocks.append(el)
    per_rank = None
    if dist is not None and world > 1:
        mine = torch.tensor([args.steps / float(np.median(own_blocks)), float(info["n_voxels"])], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"scenes_per_s": [round(float(x[0]), 2) for x in allr], "voxels": [int(x[1]) for x in allr]}
        per_rank["min"], per_rank["max"] = min(per_rank["scenes_per_s"]), max(per_rank["scenes_per_s"])
    phase("warmup_and_timed_blocks")
    elapsed = float(np.median(blocks))
    rate = lambda e: world * args.steps / e
    ret = runner.result()
    n_prop = int(ret["proposals"][1].shape[0] - 1)

    roof = cpu = single_ms = stages = grouping = tta = None
    legs = {}
    if rank == 0:
        n_probe = max(2, min(args.steps, 5)) * args.inflight

        def probe_leg(run, n):
            probe = ConvProbe()
            probe.install()
            run(n)
            n_launch, t_ms, nbytes, flops = probe.summary()
            probe.remove()
            achieved = nbytes / (t_ms * 1e-3) / 1e9
            leg = {"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                   "launches_per_step": n_launch // n, "avg_launch_us": round(t_ms * 1e3 / n_launch, 2),
                   "algorithmic_bytes_per_launch": int(nbytes / n_launch),
                   "algorithmic_bytes_per_launch_r01_accounting": int(probe.nbytes_r1 / n_launch),
                   "achieved_tflops": round(flops / (t_ms * 1e-3) / 1e12, 2), "by_level": []}
            for stride in sorted(probe.levels):
                c, ms, by, fl = probe.levels[stride]
                gbs = by / (ms * 1e-3) / 1e9
                leg["by_level"].append({"tensor_stride": stride, "launches_per_step": c // n, "avg_launch_us": round(ms * 1e3 / c, 2),
                                        "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                                        "achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 2)})
            # round 6: the same split per kernel FAMILY (which family pbn_spconv_forward's automatic choice gives each op), and
            # against the SQ counters of the newest committed profile: MFMAs issued (busy cycles / 16) over the MFMAs the
            # launches' rule pairs need (flops / 16 384 per 16x16x32 MFMA; true channel counts)
            sq = sq_summary()
            leg["by_family"] = []
            for fam in sorted(probe.families):
                c, ms, by, fl = probe.families[fam]
                gbs = by / (ms * 1e-3) / 1e9
                row = {"family": fam, "launches_per_step": c // n, "avg_launch_us": round(ms * 1e3 / c, 2), "us_per_step": round(ms * 1e3 / n, 1),
                       "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 2)}
                key = {"workgroup-tile (k_spconv)": ("tile_family_wide", "tile_family_coarse"), "wave-autonomous (k_spconv_wave)": ("wave_family",),
                       "row-stationary (k_spconv_rs)": ("row_stationary", "row_stationary_gather")}.get(fam, ())
                if sq is not None:
                    gs = [sq[0][k_] for k_ in key if k_ in sq[0] and "mfma_busy_cycles_per_launch" in sq[0][k_]]
                    if gs:
                        busy = sum(g_["mfma_busy_cycles_per_launch"] * g_["launches"] for g_ in gs) / sum(g_["launches"] for g_ in gs)
                        row["mfma_issued_over_useful"] = round((busy / 16.0) / max(fl / c / 16384.0, 1.0), 2)
                        row["mfma_source"] = sq[1]
                leg["by_family"].append(row)
            return leg

        # the launches are timed in the mode the timed region ran in: with several scenes in flight a launch shares the
        # CUs with the other streams' kernels, so its own duration grows while the whole-job rate rises
        # (a planned / graph run is probed through the size-exact launches of the same scene: the per-op events live in the
        # size-exact executor entry; the planned forward runs the same kernels over capacity-sized buffers)
        probe_runner = runner if args.forward_mode == "size-exact" else Runner(model, b, t, args.inflight, device)
        leg = probe_leg(probe_runner.run, n_probe)
        traffic, traffic_src = pmc_traffic(args)
        roof = {"bound": "hbm", "achieved": leg.pop("achieved"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": leg.pop("frac"), "traffic": traffic, "traffic_source": traffic_src, "kernel": "k_spconv*"}
        roof.update(leg)
        # the same bytes against the wall clock of the whole job (SURVEY.md 8d's path-level form, convolution bytes only):
        # what the GPU sustains across all in-flight scenes, non-convolution stages included in the time
        conv_bytes_per_scene = roof["algorithmic_bytes_per_launch"] * roof["launches_per_step"]
        path_gbs = conv_bytes_per_scene * args.steps / elapsed / 1e9
        roof["path_level"] = {"conv_algorithmic_bytes_per_scene": conv_bytes_per_scene, "achieved": round(path_gbs, 1),
                              "frac": round(path_gbs / HBM_PEAK_GBS, 4)}

        def one_at_a_time(n):
            for _ in range(n):
                one_step(model, b, t)
        if args.inflight > 1:
            roof["one_scene_in_flight"] = probe_leg(one_at_a_time, max(2, min(args.steps, 5)))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_at_a_time(10)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / 10 * 1e3
        phase("roofline_probes")
        if not args.no_extras:
            # per-stage wall time, one scene alone on the GPU, device synchronised at every stage boundary
            from pbnet_amd import prof
            prof.reset()
            prof.enable(True)
            one_at_a_time(5)
            prof.enable(False)
            rep = prof.report()
            g = lambda *names: round(sum(rep[k][0] for k in names if k in rep), 3)
            stages = {"coords_and_maps_backbone": g("a3_coords"), "unet_backbone_34C": g("a4_unet"),
                      "heads_and_gathers": g("a5_heads_gather"), "class_selection": g("a6_select"),
                      "grouping": g("a7_16_grouping"), "local_scene_plan_host": g("a17_plan"),
                      "local_scene_rows": g("a17_gather"), "coords_and_maps_mask": g("a18_mask_coords"),
                      "unet_mask_14A": g("a18_mask_unet"), "proposals": g("a19_proposals"),
                      "coords_and_maps_score": g("a20_score_coords"), "unet_score_34C": g("a20_score_unet"),
                      "pool_and_score_head": g("a20_pool_head")}
            stages["sum"] = round(sum(stages.values()), 3)
            # voxelisation happens before the path (inputs arrive voxelised); timed here for SURVEY 8d's list
            from pbnet_amd import loader_ops
            xyz = [b["xyz_original"].double()]
            ft = [torch.zeros(b["xyz_original"].shape[0], 6, device=device)]
            loader_ops.voxelize_batch(xyz, ft, WORKLOADS[args.workload]["voxel"])
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(5):
                loader_ops.voxelize_batch(xyz, ft, WORKLOADS[args.workload]["voxel"])
            torch.cuda.synchronize()
            stages["voxelise_outside_the_metric"] = round((time.perf_counter() - tv) / 5 * 1e3, 3)
            m = grouped_points(raw)
            grouping = {"points_grouped_per_forward": m, "us": round(stages["grouping"] * 1e3, 1),
                        "points_per_s": round(m / (stages["grouping"] * 1e-3), 0) if stages["grouping"] > 0 else None,
                        "note": "stage wall time incl. its read-back of the cluster table, one scene alone on the GPU"}
            phase("stage_breakdown")
            if args.copies == 1 and world == 1:
                # the reference's eval batch: 3 rotated copies of the scene per forward (dataset_preprocess.py:324)
                b3, t3, info3, _ = build_scene(3, dtype, device, args.workload)
                r3 = Runner(model, b3, t3, args.inflight, device)
                r3.run(args.inflight)
                r3.run(max(2, args.warmup // 2))
                k3 = max(args.inflight, args.steps // 3)
                bl3 = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r3.run(k3)
                    torch.cuda.synchronize()
                    bl3.append(time.perf_counter() - t0)
                e3 = float(np.median(bl3))
                tta = {"value": round(k3 / e3, 3), "unit": "scenes/s (1 scene = 3 rotated copies per forward)",
                       "copies_per_s": round(3 * k3 / e3, 1), "ms_per_step": round(e3 / k3 * 1e3, 3), "steps": k3,
                       "points_per_step": info3["n_points"], "voxels_per_step": info3["n_voxels"],
                       "proposals_per_step": int(r3.result()["proposals"][1].shape[0] - 1)}
                del b3, t3, r3
                phase("tta3_leg")
            if args.copies == 1 and world == 1 and args.workload == "c2":
                # the other configurations of BASELINE.json on the same line (each a short run; the headline stays configs[1]):
                # fp32 = the parity configuration (1e-4 against the oracle is asserted in fp32: tests/test_bench_workload_gpu.py)
                k = max(args.inflight, args.steps // 3)
                b32 = dict(b, feat_voxel=b["feat_voxel"].float())
                t32 = {k_: torch.from_numpy(v_).to(device) for k_, v_ in raw[1].items()}     # the teacher in fp32 too (not the 16-bit copy)
                v, ms, alone, _ = timed_leg(model, b32, t32, args.inflight, device, k)
                legs["fp32"] = {"value": round(v, 3), "unit": "scenes/s", "ms_per_step": round(ms, 3),
                                "one_scene_in_flight_ms_per_scene": round(alone, 3), "dtype": "f32",
                                "note": "configs[1] with fp32 feature slabs (v_mfma_f32_16x16x4_f32): the configuration the "
                                        "1e-4 parity tests run in"}
                del b32, t32
                # configs[4]: fp16 feature slabs, int32 coordinates, the WHOLE forward replayed from a HIP graph (planned.py), one
                # graph per stream; and the same sync-free forward in the headline's dtype, eager and from graphs: the distance
                # to the headline is the host's share (five read-backs + the Python of the eager forward)
                legs["graph_f16"] = planned_leg(model, b, t, torch.float16, args.inflight, device, k, graph=True)
                legs["graph_f16"]["note"] = ("configs[4]: fp16 slabs + int32 coordinates, MinkUNet34C, whole PBNet.forward from a HIP "
                                             "graph per stream (capacity-planned forward, no host read-back inside)")
                legs["planned"] = {"eager": planned_leg(model, b, t, dtype, args.inflight, device, k, graph=False),
                                   "graph": planned_leg(model, b, t, dtype, args.inflight, device, k, graph=True),
                                   "note": "the sync-free forward in the headline's dtype; headline = eager PBNet.forward with its "
                                           "read-backs"}
                phase("planned_legs")
                # three DISTINCT scenes per forward through the batch index: every launch of the coarse levels gets 3x the rows
                bb, tb, ib = build_batched((2, 4, 5), dtype, device)
                v, ms, alone, rb = timed_leg(model, bb, tb, args.inflight, device, max(args.inflight, k // 2))
                legs["batched3"] = {"value": round(3 * v, 3), "unit": "scenes/s (3 distinct scenes per forward)",
                                    "forwards_per_s": round(v, 3), "ms_per_forward": round(ms, 3),
                                    "one_forward_in_flight_ms": round(alone, 3), "points_per_forward": ib["n_points"],
                                    "voxels_per_forward": ib["n_voxels"],
                                    "proposals_per_forward": int(rb["proposals"][1].shape[0] - 1)}
                del bb, tb, rb
                # round 6: the serving front (pbnet_amd/serving.py) -- the scenes waiting on the GPU merged into one forward through
                # the reference's batch axis, B scenes per forward x F forwards in flight, per-scene results split afterwards;
                # a stream of DISTINCT configs[1] scenes (seeds 2..9: 150-180 k points each)
                legs["served"] = served_legs(model, dtype, device, args.inflight)
                phase("served_legs")
                # configs[3]: the dense 1 cm scene (rulebook build + gather/scatter stress)
                b4, t4, i4, _ = build_scene(1, dtype, device, "c4")
                v, ms, alone, r4 = timed_leg(model, b4, t4, args.inflight, device, max(args.inflight, k // 4), warmup=args.inflight)
                legs["c4"] = {"value": round(v, 3), "unit": "scenes/s", "ms_per_step": round(ms, 3),
                              "one_scene_in_flight_ms_per_scene": round(alone, 3), "points_per_step": i4["n_points"],
                              "voxels_per_step": i4["n_voxels"], "proposals_per_step": int(r4["proposals"][1].shape[0] - 1),
                              "workload": "configs[3]: %d pts, %d voxels @1cm" % (i4["n_points"], i4["n_voxels"])}
                del b4, t4, r4
                torch.cuda.empty_cache()
                phase("config_legs")
                # configs[2] on this one rank: model_fn forward + losses + backward + gradient reduction + Adam
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import train_step as TS
                ph = {}
                e, comm, loss, it = TS.run_training(world, rank, device, dist, steps=12, warmup=3, phases_out=ph)
                legs["train_step"] = {"value": round(12 / e, 3), "unit": "scenes/s per rank (configs[2]: bf16 training step, "
                                      "one scene per rank)", "ms_per_step": round(e / 12 * 1e3, 2),
                                      "allreduce_tail_ms_per_step": round(comm * 1e3, 2), "loss": round(loss, 5),
                                      "points_per_scene": it["n_points"], "voxels_per_scene": it["n_voxels"],
                                      "phases_ms_synchronised": {k_: round(v_, 2) for k_, v_ in ph.items()},
                                      "roofline": it.get("roofline")}
                phase("train_step_leg")
            if world == 1 and not args.no_cpu_baseline:
                cpu = cpu_baseline(cfg, model, raw)
                phase("cpu_baseline")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        phases["total"] = round(time.perf_counter() - T_START, 2)
        line = {
            "metric": "scenes/sec fwd+cluster (ScanNet ~150k pts/scene)",
            "value": round(rate(elapsed), 3),
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
            "gpu_active": gpu_active(args),
            "timed_blocks": {"n": len(blocks), "steps_per_block": args.steps, "timed_seconds": round(float(sum(blocks)), 3),
                             "cv": round(float(np.std(blocks) / np.mean(blocks)), 4),
                             "statistic": "median", "value_p10": round(rate(float(np.percentile(blocks, 90))), 3),
                             "value_p90": round(rate(float(np.percentile(blocks, 10))), 3),
                             "ms_per_step_all": [round(e / args.steps * 1e3, 3) for e in blocks]},
            "config": {"workload": "%s: 1 scene, %d pts, %d voxels @%gcm, %d rotated cop%s, full "
                                   "PBNet.forward (MinkUNet34C + grouping + MinkUNet14A mask + MinkUNet34C score)"
                                   % ("configs[1]" if args.workload == "c2" else "configs[3]",
                                      info["n_points"] // args.copies, info["n_voxels"] // args.copies,
                                      WORKLOADS[args.workload]["voxel"] * 100, args.copies,
                                      "y" if args.copies == 1 else "ies"),
                       "forward_path": "size-exact" if args.forward_mode == "size-exact" else args.forward_mode + " (pbnet_amd/planned.py, capacities 1.25 x this scene's sizes; NOT the headline configuration)",
                       "points_per_step": info["n_points"], "voxels_per_step": info["n_voxels"],
                       "proposals_per_step": n_prop, "scenes_in_flight_per_gpu": args.inflight,
                       "one_scene_in_flight_ms_per_scene": None if single_ms is None else round(single_ms, 3),
                       "parallelism": "scenes sharded over GPUs, %d in flight per GPU (host thread + HIP stream each), "
                                      "no data-path collective" % args.inflight,
                       "scene_seeds": "2 (configs[1])" if world == 1 else "10 + rank (SURVEY C3/C5)",
                       "cpu_pinning": pinned,
                       "rccl": rccl_status},
            "roofline": roof,
        }
        if per_rank is not None:
            line["per_rank"] = per_rank
        if stages is not None:
            line["stages_ms"] = stages
            line["grouping"] = grouping
        if tta is not None:
            line["tta3"] = tta
        line.update(legs)
        if cpu is not None:
            line["cpu_baseline"] = cpu
        line["phases_s"] = phases
        # RCCL prints its version banner through C stdio, which is flushed at exit when stdout is a file: flush it now so
        # that the JSON line is the LAST line of stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
