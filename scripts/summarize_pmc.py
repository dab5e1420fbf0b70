#!/usr/bin/env python3
"""Aggregate the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
into per-kernel-family averages and the per-launch HBM traffic of the dominant kernel.

    python scripts/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_summary.json

Units / corrections (guide, section HBM): FETCH_SIZE and WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads (16 B per lane: the gather and LDS-DMA loads of k_spconv), so it is doubled.  WRITE_SIZE is
used as reported (uncalibrated on gfx950).  Infinity-Cache hits are counted by both."""
import collections
import csv
import json
import sys


def family(name):
    if "k_spconv_reduce" in name:
        return "k_spconv_reduce"
    if "k_spconv<" in name or "k_spconv_wave<" in name or "k_spconv_rs<" in name or "k_spconv_rsh<" in name:      # every convolution kernel family
        return "k_spconv"
    if "pbn::" in name:
        return "pbn_other"
    return "runtime_torch"


def load(d):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(d + "/bench_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            a = agg[family(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch), load(write)
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python bench.py --no-extras --steps 5 --warmup 2 --repeats 1",
           "units": "bytes per launch; FETCH_SIZE x2 (gfx950 wide-load correction), WRITE_SIZE as reported, both x1024 (KB)",
           "families": {}}
    for k in sorted(set(f) | set(w)):
        fb = 2.0 * 1024.0 * f[k][1] / max(f[k][0], 1)
        wb = 1024.0 * w[k][1] / max(w[k][0], 1)
        res["families"][k] = {"launches": f[k][0], "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                              "traffic_bytes_per_launch": round(fb + wb)}
    conv = res["families"]["k_spconv"]
    res["k_spconv_traffic_bytes_per_launch"] = conv["traffic_bytes_per_launch"]
    red = res["families"].get("k_spconv_reduce")
    # per convolution OP: its own launch + its share of the split-K reduce launches
    extra = red["traffic_bytes_per_launch"] * red["launches"] / max(conv["launches"], 1) if red else 0
    res["reduce_launches_per_conv_launch"] = round(red["launches"] / max(conv["launches"], 1), 4) if red else 0.0
    res["traffic_bytes_per_conv_op_incl_reduce"] = round(conv["traffic_bytes_per_launch"] + extra)
    with open(out, "w") as fo:
        json.dump(res, fo, indent=1)
    print(json.dumps(res["families"], indent=1))


if __name__ == "__main__":
    main()
