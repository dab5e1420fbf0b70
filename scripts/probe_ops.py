"""Per-op durations of the three native U-Net executors inside one bench step (HIP events per op)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pbnet_amd.network import mink_unet as U

cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, torch.device("cuda", 0))
for _ in range(3):
    bench.one_step(model, b, t)
acc = {}
calls = []
KN = {0: "1x1", 1: "k3", 2: "k5", 3: "down", 4: "up"}
def sink(m, plan, rows, cm, esz, op_ms):
    calls.append((type(m).__name__, rows, [(plan["ops"][i].map_kind, plan["ops"][i].level_in, plan["ops"][i].level_out,
                                            plan["ops"][i].vpo * (16 // esz), plan["ops"][i].cout_p, op_ms[i]) for i in range(plan["n_ops"])]))
U.MinkUNet.OP_TIMING_SINK = sink
REP = 5
for _ in range(REP):
    bench.one_step(model, b, t)
U.MinkUNet.OP_TIMING_SINK = None
nets = len(calls) // REP
for ni in range(nets):
    name, rows, _ = calls[ni]
    tot = 0.0
    agg = {}
    for r in range(REP):
        for (kind, lin, lout, cin, cout, ms) in calls[r * nets + ni][2]:
            key = (KN[kind], lin, lout, cin, cout)
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1; a[1] += ms
    print("== net %d rows %s" % (ni, rows))
    for key, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-5s L%d->L%d %3d->%3d  x%d  %.1f us each, %.3f ms/step" % (key + (cnt // REP, ms / cnt * 1e3, ms / REP)))
        tot += ms / REP
    print("  total %.3f ms/step" % tot)
