"""Concurrency picture of a rocprofv3 kernel trace (several scenes in flight): busy fraction of the wall window, time-weighted
number of kernels running, and per kernel family the time it ran alone / its share of the 'kernel-seconds'.
usage: analyze_trace.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
def fill(r):
    """Rough share of the 256 CUs a launch can keep busy: workgroups / (CUs x resident workgroups per CU by LDS/waves)."""
    try:
        wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]); lds = int(r.get("LDS_Block_Size", 0) or 0)
    except (KeyError, ValueError):
        return 1.0
    n_wg = max(1, grid // wg)
    per_cu = max(1, min(32 * 64 // wg, (160 * 1024) // lds if lds > 0 else 32))
    return min(1.0, n_wg / (256.0 * per_cu))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], fill(r)) for r in rows]
ev.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip
t_hi = ev[0][0] + (ev[-1][1] - ev[0][0]) * 0.80
ev = [e for e in ev if e[0] >= t_lo and e[1] <= t_hi]
def fam(n):
    m = re.search(r"(k_\w+|Cijk\w{0,12}|\w+)", n.replace("void ", "").replace("pbn::(anonymous namespace)::", ""))
    n2 = m.group(1) if m else n[:30]
    if n2.startswith("k_spconv") and "reduce" not in n2:
        t = re.search(r"k_spconv<[^,]+, (\d+), (\d+)", n)
        n2 = "k_spconv<NT=%s>" % (t.group(2) if t else "?")
    return n2[:40]
pts = []
fills = []
for s, e, n, f in ev:
    pts.append((s, 1, fam(n))); pts.append((e, -1, fam(n)))
    fills.append((s, f)); fills.append((e, -f))
fills.sort()
cur = 0.0; lastt = fills[0][0]; area = 0.0; capped = 0.0
for t, d in fills:
    area += cur * (t - lastt); capped += min(cur, 1.0) * (t - lastt); lastt = t; cur += d
print("sum of launch fill factors over the window: mean %.2f GPUs demanded, %.1f%% of the CU capacity coverable" % (area / (fills[-1][0] - fills[0][0]), 100 * capped / (fills[-1][0] - fills[0][0])))
pts.sort(key=lambda p: (p[0], p[1]))
running = collections.Counter(); nrun = 0
hist = collections.Counter(); alone = collections.Counter(); ksec = collections.Counter()
last = pts[0][0]
for t, d, n in pts:
    dt = t - last
    if dt > 0:
        hist[min(nrun, 8)] += dt
        for k, c in running.items():
            if c > 0:
                ksec[k] += dt * c
                if nrun == 1:
                    alone[k] += dt
    last = t
    running[n] += d; nrun += d
wall = pts[-1][0] - pts[0][0]
print("window %.1f ms, %d kernels; busy %.1f%%" % (wall / 1e6, len(ev), 100 * (1 - hist[0] / wall)))
print("concurrency (share of wall): " + "  ".join("%d:%.1f%%" % (k, 100 * v / wall) for k, v in sorted(hist.items())))
tot = sum(ksec.values())
print("%-42s %8s %8s %8s" % ("family", "k-sec %", "alone ms", "alone %wall"))
for k, v in ksec.most_common(22):
    print("%-42s %7.1f%% %8.2f %7.1f%%" % (k, 100 * v / tot, alone[k] / 1e6, 100 * alone[k] / wall))
# machine-readable summary next to the text (bench.py reports gpu_active_frac from the newest committed one)
import json, os
out_json = os.environ.get("PBN_TRACE_JSON")
if out_json:
    with open(out_json, "w") as fh:
        json.dump({"what": "union of kernel intervals / wall over the middle of a rocprofv3 --kernel-trace run of bench.py (profiler attached: "
                           "the run is slower than the plain one)", "window_ms": round(wall / 1e6, 2), "kernels": len(ev),
                   "gpu_active_frac": round(1 - hist[0] / wall, 4),
                   "concurrency_share_of_wall": {str(k): round(v / wall, 4) for k, v in sorted(hist.items())},
                   "mean_kernels_running": round(sum(k * v for k, v in hist.items()) / wall, 3)}, fh, indent=1)
