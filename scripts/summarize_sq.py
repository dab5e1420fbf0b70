#!/usr/bin/env python3
"""SQ counters of the convolution kernels (one rocprofv3 --pmc pass + the kernel trace of the same run):
    python scripts/summarize_sq.py <counter_collection.csv> <kernel_trace.csv> <out.json>
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs) summed over the launches of each group."""
import collections, csv, json, sys

cc, kt, out = sys.argv[1:4]
dur = {}
try:
    with open(kt) as f:
        for r in csv.DictReader(f):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3     # us
except OSError:
    pass        # the counter file carries the timestamps of every dispatch as well


def group(name, grid):
    if "k_spconv_reduce" in name:
        return "reduce"
    if "k_spconv_wave<" in name:
        return "wave_family"
    if "k_spconv_rs<" in name:
        return "row_stationary"
    if "k_spconv<" in name:
        return "tile_family_wide" if grid >= 400 * 256 else "tile_family_coarse"
    return None


agg = collections.defaultdict(lambda: {"launches": set(), "us": 0.0, "c": collections.Counter()})
with open(cc) as f:
    for r in csv.DictReader(f):
        g = group(r["Kernel_Name"], int(r.get("Grid_Size", "0") or 0))
        if g is None:
            continue
        a = agg[g]
        if r["Dispatch_Id"] not in a["launches"]:
            a["launches"].add(r["Dispatch_Id"])
            a["us"] += dur.get(r["Dispatch_Id"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
        a["c"][r["Counter_Name"]] += float(r["Counter_Value"])
res = {"source": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                 "(own pass) -- python bench.py --no-extras --inflight 1 --steps 5 --warmup 2 --repeats 1",
       "groups": {}}
for g, a in sorted(agg.items()):
    c = a["c"]
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    res["groups"][g] = {"launches": len(a["launches"]), "sum_duration_us": round(a["us"], 1),
                        "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0) / wc, 4),
                        "wait_inst_frac": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4),
                        "active_inst_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4),
                        "mfma_busy_over_simd_cycles_at_2.4GHz": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(a["us"] * 2400.0 * 1024.0, 1.0), 4),
                        # round 6: the counter itself, per launch -- 16 busy cycles per v_mfma_f32_16x16x32_{bf16,f16} (8 passes of 2
                        # cycles... measured: 17 cycles back to back on one SIMD, MI355X_MICROARCH.md), so busy / 16 = MFMAs ISSUED;
                        # bench.py divides by the MFMAs the launches' rule pairs need (roofline.by_family[].mfma_issued_over_useful)
                        "mfma_busy_cycles_per_launch": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(len(a["launches"]), 1), 1)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["groups"], indent=1))
