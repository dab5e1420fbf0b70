"""CPU tier: the N-rank ENTRY POINTS themselves under the driver's launcher (`python -m torch.distributed.run --nnodes=1
--nproc-per-node 2 --master-addr 127.0.0.1 --master-port P <entry> --gpus 2 ...`), with gloo standing in for RCCL and a stand-in
step for the device work (`--dry-run`): argument parsing, environment rendezvous, barriers, MAX-over-ranks timing, scene
sharding and the ONE JSON line of rank 0 are exercised before an 8-GPU node ever runs them.  No scaling curve has been
measured on hardware: this pins the plumbing only (/root/reference/train.py:323-345, eval_map.py:48-50)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(entry, extra, nproc=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, entry)] + extra
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # ONE JSON line, printed by rank 0 only
    return json.loads(lines[0])


def test_bench_entry_two_ranks():
    line = _launch("bench.py", ["--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["unit"] == "scenes/s"
    assert abs(line["value"] - 2 * 4 / (line["ms_per_step"] * 4e-3)) / line["value"] < 1e-3     # whole-job aggregate over both ranks
    assert line["config"]["scene_shard_sizes"] == [8, 8]                                       # 16 scenes, round-robin
    assert line["config"]["rccl"]["selfcheck"]["ranks_seen"] == 2 and line["config"]["rccl"]["selfcheck"]["all_ok"]
    assert line["roofline"] is None


def test_bench_entry_refuses_a_mismatched_world():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0


def test_train_step_entry_two_ranks():
    line = _launch("scripts/train_step.py", ["--steps", "3", "--warmup", "1", "--dry-run"])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["steps"] == 3
    assert line["replicas_identical"] is True            # the same averaged gradient reached both replicas every step
    assert line["unused_grads_none"] is True             # find_unused_parameters semantics: no state for untrained branches


def test_bench_spawns_its_own_ranks_eight_ranks_four_threads():
    """`python bench.py --gpus 8` (the N = 1 command with another N, no launcher, WORLD_SIZE unset) starts
    its eight ranks itself -- children of a process that never touches the GPU -- and rank 0's ONE JSON line comes back; every rank
    runs its steps on four host threads (the in-flight mode), i.e. 8 x 4 threads on this box's cores: the host-side shape of the
    8-GPU run, with gloo and a stand-in step."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "16", "--warmup", "4",
                          "--inflight", "4", "--dry-run"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["dry_run"] is True and line["n_gpus"] == 8 and line["steps"] == 16
    assert line["config"]["scene_shard_sizes"] == [8] * 8 and line["config"]["host_threads_per_rank"] == 4
    chk = line["config"]["rccl"]["selfcheck"]              # the rank-stamped bucket all-reduce every rank verifies (RCCL on the GPU run)
    assert chk["ranks_seen"] == 8 and chk["all_ok"] and [r["rank"] for r in chk["per_rank"]] == list(range(8))

