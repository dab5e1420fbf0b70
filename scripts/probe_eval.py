"""Evaluator association (tools/eval.py:205-250) on a full-size scene: device overlap table vs the oracle's per-pair passes
(which is how the reference counts).  Manual timing script; the oracle is the checker."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np, torch
from tests.test_eval_gpu import _big_scene, _flat, _sorted_rows
from pbnet_amd import evaluate as E
from oracle import evaluate_ref as O
gt, pred = _big_scene(1, 161517, 60, 48)
dev_pred = dict(pred, mask=torch.from_numpy(np.ascontiguousarray(pred["mask"])).to("cuda:0"))
for _ in range(3):
    rec = E.assign_instances_for_scan("big", dev_pred, gt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    rec = E.assign_instances_for_scan("big", dev_pred, gt)
t_dev = (time.perf_counter() - t0) / 20
idx = torch.from_numpy(np.unique(gt, return_inverse=True)[1].astype(np.int32)).to("cuda:0")
inter = torch.empty(48, int(idx.max()) + 1, dtype=torch.int32, device="cuda:0")
from pbnet_amd import _native as N
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    N.lib().pbn_instance_overlap(N.ptr(dev_pred["mask"]), 48, gt.shape[0], N.ptr(idx), int(idx.max()) + 1, N.ptr(inter), N.current_stream())
e1.record(); torch.cuda.synchronize()
t_k = e0.elapsed_time(e1) / 50
t0 = time.perf_counter()
want = O.assign(pred["conf"], pred["label_id"], pred["mask"], gt)
t_cpu = time.perf_counter() - t0
got = _flat(rec)
ok = all(np.array_equal(a, b) for a, b in zip(got[:2], want[:2])) and np.array_equal(got[3], _sorted_rows(want[3]))
mb = 48 * gt.shape[0] * 4 / 1e6
print("48 predictions x %d points, %d instances: association %.2f ms (kernel+memset %.1f us = %.0f GB/s of mask bytes), "
      "per-pair numpy %.0f ms, identical=%s" % (gt.shape[0], want[0].shape[0], t_dev * 1e3, t_k * 1e3, mb / t_k, t_cpu * 1e3, ok))
