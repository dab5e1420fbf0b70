#!/usr/bin/env python3
"""HBM traffic (PMC) of every convolution op of a forward next to its algorithmic bytes: which ops re-read.

    python scripts/pmc_by_op.py <fetch_dir> <write_dir> <op_table.txt>

The two directories hold bench_counter_collection.csv of a `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` run of
`bench.py --no-extras --inflight 1 ...` (one scene in flight: the convolution dispatches of a forward arrive in op order, 138 per
forward); op_table.txt is scripts/op_table.py's output for the same scene (algorithmic MB per op).  FETCH_SIZE x2 x1024 and
WRITE_SIZE x1024 as in scripts/summarize_pmc.py."""
import collections
import csv
import re
import sys


def conv_dispatches(d):
    rows = []
    with open(d + "/bench_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "k_spconv" in n and "reduce" not in n:
                rows.append((int(r["Dispatch_Id"]), n, float(r["Counter_Value"])))
    rows.sort()
    return rows


def main():
    fetch, write, table = sys.argv[1:4]
    ops = []
    for line in open(table):
        m = re.match(r"^(\S*)\s+(\d+)\s+(\d+)\s+(\d+)>\s*(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m and not line.startswith("net"):
            ops.append({"op": int(m.group(2)), "lo": int(m.group(5)), "v_out": int(m.group(7)), "cin": int(m.group(8)), "cout": int(m.group(9)),
                        "K": int(m.group(10)), "mb": float(m.group(12)), "us": float(m.group(14))})
    n_ops = len(ops)
    f, w = conv_dispatches(fetch), conv_dispatches(write)
    nf = len(f) // n_ops
    assert nf >= 1 and len(f) % n_ops == 0 and len(w) == len(f), (len(f), len(w), n_ops)
    acc = [[0.0, 0.0, ""] for _ in range(n_ops)]
    for i, ((_, name, fv), (_, _, wv)) in enumerate(zip(f, w)):
        a = acc[i % n_ops]
        a[0] += 2.0 * 1024.0 * fv / nf
        a[1] += 1024.0 * wv / nf
        k = re.search(r"(k_spconv\w*<[^(]*>)\(", name)
        a[2] = (k.group(1) if k else name)[:44].replace("__hip_bfloat16", "bf16")
    print("%d forwards of %d convolution ops; traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB counters)" % (nf, n_ops))
    print("%3s %2s %7s %4s %4s %3s %9s %9s %9s %6s  %s" % ("op", "lo", "v_out", "cin", "cout", "K", "algo MB", "fetch MB", "write MB", "ratio", "kernel"))
    lev = collections.defaultdict(lambda: [0.0, 0.0])
    ta = tt = 0.0
    for o, a in zip(ops, acc):
        t = (a[0] + a[1]) / 1e6
        print("%3d %2d %7d %4d %4d %3d %9.3f %9.3f %9.3f %6.2f  %s" % (o["op"], o["lo"], o["v_out"], o["cin"], o["cout"], o["K"], o["mb"],
                                                                     a[0] / 1e6, a[1] / 1e6, t / o["mb"], a[2]))
        lev[o["lo"]][0] += o["mb"]; lev[o["lo"]][1] += t
        ta += o["mb"]; tt += t
    print("total: algorithmic %.1f MB, traffic %.1f MB, ratio %.2f" % (ta, tt, tt / ta))
    for l in sorted(lev):
        print("  out level %d: algorithmic %.1f MB, traffic %.1f MB, ratio %.2f" % (l, lev[l][0], lev[l][1], lev[l][1] / lev[l][0]))


if __name__ == "__main__":
    main()
