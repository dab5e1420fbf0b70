import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(3): bench.one_step(model, b, t)
torch.cuda.synchronize()
for i in range(30):
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    bench.one_step(model, b, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st1 = torch.cuda.memory_stats()
    print("step %2d %.2f ms  reserved=%.0f MB  new_device_allocs=%d gc=%s" % (i, dt, torch.cuda.memory_reserved() / 2**20,
          st1["num_device_alloc"] - st0["num_device_alloc"], gc.get_count()))
