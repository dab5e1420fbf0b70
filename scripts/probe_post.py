"""Post-processing (eval_map.py:55-123) on the 3-copy bench scene's own proposals: device path vs the numpy oracle."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np, torch
import bench
from pbnet_amd import postprocess as PP
from oracle import postprocess_ref as R
dev = torch.device("cuda", 0)
cfg, model, b, t, info, raw = bench.build_workload(0, 3, torch.bfloat16, dev)
ret = bench.one_step(model, b, t)
point_num = int(b["xyz_original"].shape[0])
n_fold = point_num // 3
rng = np.random.default_rng(0)
sp = np.repeat(np.arange(n_fold // 90 + 1), 90)[:n_fold].astype(np.int64)       # compressed superpoint ids, runs of 90 points
c = types.SimpleNamespace(TEST_SCORE_THRESH=-1.0, TEST_NPOINT_THRESH=101, TEST_NMS_THRESH=0.10)   # random-init scores: keep all
pred_sem = ret["sem_pred_p"]
for _ in range(3):
    out = PP.refine_instances(pred_sem, ret["proposals"], ret["clt_scores"], point_num, sp, c)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = PP.refine_instances(pred_sem, ret["proposals"], ret["clt_scores"], point_num, sp, c)
torch.cuda.synchronize()
t_gpu = (time.perf_counter() - t0) / 20
pi, po = ret["proposals"][0].cpu().numpy(), ret["proposals"][1].cpu().numpy()
t0 = time.perf_counter()
ref = R.refine_instances(pred_sem.cpu().numpy(), pi, po, ret["clt_scores"].float().cpu().numpy().reshape(-1), point_num, sp,
                         c.TEST_SCORE_THRESH, c.TEST_NPOINT_THRESH, c.TEST_NMS_THRESH)
t_cpu = time.perf_counter() - t0
ok = np.array_equal(out[0].cpu().numpy(), ref["clusters"]) and np.array_equal(out[2].cpu().numpy(), ref["cluster_semantic_id"])
print("proposals %d, points/copy %d, clusters %d: device path %.3f ms, numpy oracle %.1f ms, identical=%s"
      % (po.shape[0] - 1, n_fold, out[0].shape[0], t_gpu * 1e3, t_cpu * 1e3, ok))
