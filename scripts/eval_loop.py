"""The reference's validation loop (eval_map.py:40-151) on this package, end to end, on synthetic scenes:
loader block (voxelise + collate on the device) -> model_fn_eval (PBNet.forward) -> refine_instances (TTA fold, thresholds,
NMS, superpoint vote) -> assign_instances_for_scan -> [ranks merge] -> evaluate_matches / compute_averages / print_results.
Weights are random (no checkpoint is reachable offline) and the heads are teacher-forced, so the AP it prints measures
nothing -- the script shows the call sequence a maintainer would write and times its stages.

    python scripts/eval_loop.py [n_scenes=4]          (N ranks: torchrun --nproc-per-node N scripts/eval_loop.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pbnet_amd import dist as pdist, evaluate, loader_ops, synth
from pbnet_amd.config import get_config
from pbnet_amd.network.PBNet import PBNet, model_fn_eval
from pbnet_amd.postprocess import refine_instances

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
if world > 1:
    torch.distributed.init_process_group("nccl", device_id=dev)
cfg = get_config(test=True)
torch.manual_seed(22)
model = PBNet(cfg).to(dev).eval()
timing, matches = {}, {}
def lap(name, t0):
    torch.cuda.synchronize(); timing[name] = timing.get(name, 0.0) + time.perf_counter() - t0
for scene in pdist.shard_scenes(n_scenes, rank, world):
    name = "scene%04d_00" % scene
    raw, teacher, _ = synth.make_val_batch(seed=20 + scene, copies=1, room=(2.4, 2.0, 1.8), n_boxes=6)
    n = raw["xyz_original"].shape[0]
    rot = [np.eye(3), np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]), np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1.0]])]
    xyz3 = [raw["xyz_original"] @ r.T.astype(np.float32) for r in rot]                 # the 3 TTA copies (dataset_preprocess.py:324)
    feat3 = [raw["feat_voxel"][raw["v2p_index"]]] * 3
    t0 = time.perf_counter()
    xyz_voxel, feat_voxel, v2p = loader_ops.voxelize_batch([x - x.min(0) for x in xyz3], feat3, cfg.voxel_size, dev)
    batch = dict(xyz_voxel=xyz_voxel, feat_voxel=feat_voxel.to(torch.bfloat16), v2p_index=v2p,
                 xyz_original=torch.from_numpy(np.concatenate(xyz3)).to(dev))
    lap("loader (voxelise + collate)", t0)
    t3 = {k: torch.from_numpy(np.concatenate([v] * 3)).to(dev) for k, v in teacher.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        pred = model_fn_eval(batch, model, 1, cfg, teacher=t3)
    lap("model_fn_eval (forward + grouping)", t0)
    superpoint = np.arange(n) // 64                                                    # stand-in for the mesh segmentation
    t0 = time.perf_counter()
    clusters, scores, sem_id = refine_instances(pred["sem"], pred["proposals"], pred["clt_scores"], 3 * n, superpoint, cfg)
    lap("refine_instances (fold, NMS, superpoints)", t0)
    sem = teacher["sem_score"].argmax(1)
    gt_ids = evaluate.encode_gt_ids(sem, raw["ins"])
    t0 = time.perf_counter()
    matches[name] = evaluate.assign_instances_for_scan(name, dict(conf=scores, label_id=sem_id, mask=clusters), gt_ids)
    lap("assign_instances_for_scan", t0)
    print("rank %d %s: %d points x3, %d proposals -> %d clusters" % (rank, name, n, pred["proposals"][1].shape[0] - 1, clusters.shape[0]))
merged = pdist.gather_scene_results(matches)
if rank == 0:
    t0 = time.perf_counter()
    avgs = evaluate.compute_averages(evaluate.evaluate_matches(merged))
    lap("evaluate_matches + averages (%d scenes)" % len(merged), t0)
    evaluate.print_results(avgs)
    for k, v in timing.items():
        print("%-46s %8.2f ms" % (k, v * 1e3))
if world > 1:
    torch.distributed.destroy_process_group()
