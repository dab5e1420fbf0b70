"""Which torch ops of one size-exact forward end in a memcpy / copy kernel (aten::copy_, aten::_to_copy, aten::cat, aten::clone,
aten::contiguous, aten::index ...): name, input shapes, count -- the glue between the native calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench as B

dev = torch.device("cuda", 0)
cfg, model, b, t, info, raw = B.build_workload(0, 1, torch.bfloat16, dev, "c2", 1)
for _ in range(3):
    B.one_step(model, b, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    B.one_step(model, b, t)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("aten ops with device time: %.1f us in one forward" % tot)
for r in rows[:45]:
    print("%8.1f us %3d x %-28s %s" % r)
