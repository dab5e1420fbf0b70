"""Experiment: the in-flight streams of the bench confined to CU subsets (hipExtStreamCreateWithCUMask): does a scene that
stays on two XCDs (its weights and rows in two L2s instead of eight) beat four scenes sharing all CUs?
usage: probe_cumask.py [mode ...]   modes: none | contig (stream i: CUs 64 i .. 64 i + 63) | inter (CU j -> stream (j / 8) % 4 ...)
       | half (two halves, two streams each) | xcdN (mask bit b -> XCD b % 8: stream i gets XCDs 2 i, 2 i + 1)"""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
cfg, model, b, t, info, raw = bench.build_workload(0, 1, torch.bfloat16, dev)
for _ in range(3):
    bench.one_step(model, b, t)
torch.cuda.synchronize()
N_CU, INF = 256, 4


def masked_streams(mode):
    if mode == "none":
        return [torch.cuda.Stream(dev) for _ in range(INF)]
    out = []
    for i in range(INF):
        bits = np.zeros(N_CU, dtype=bool)
        if mode == "contig":
            bits[64 * i:64 * (i + 1)] = True
        elif mode == "inter":
            bits[np.arange(N_CU) % INF == i] = True
        elif mode == "half":
            bits[128 * (i // 2):128 * (i // 2 + 1)] = True
        elif mode == "xcd":                       # if mask bit b lives on XCD b % 8: XCDs 2 i, 2 i + 1
            bits[np.isin(np.arange(N_CU) % 8, (2 * i, 2 * i + 1))] = True
        elif mode == "xcdhalf":                   # XCDs 0-3 / 4-7 under the same assumption
            bits[np.isin(np.arange(N_CU) % 8, range(4 * (i // 2), 4 * (i // 2) + 4))] = True
        else:
            raise SystemExit("unknown mode " + mode)
        words = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1)
        arr = (ctypes.c_uint32 * len(words))(*[int(w) for w in words])
        s = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
        assert rc == 0, rc
        out.append(torch.cuda.ExternalStream(s.value, device=dev))
    return out


for mode in (sys.argv[1:] or ["none", "contig", "inter", "half", "xcd", "xcdhalf", "none"]):
    r = bench.Runner(model, b, t, INF, dev)
    r.streams = masked_streams(mode)
    r.run(8)
    rates = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(40)
        torch.cuda.synchronize()
        rates.append(40 / (time.perf_counter() - t0))
    print("%-8s %s scenes/s (median %.1f)" % (mode, " ".join("%.1f" % v for v in rates), float(np.median(rates))), flush=True)
