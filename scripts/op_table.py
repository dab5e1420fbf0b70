"""Per-op table of the convolution ops of one bench scene (one scene alone on the GPU, HIP events around every op of the native
U-Net executor): network, op index, tensor strides, rows, channels, offsets, pairs, algorithmic bytes, dense and
useful flops, microseconds (median over the repeats), HBM fraction.  scripts/op_table.py [repeats] [dtype] > table"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
    dev = torch.device("cuda:0")
    cfg, model, b, t, info, raw = bench.build_workload(0, 1, dtype, dev)
    from pbnet_amd.network import mink_unet as U
    K_OF_KIND = bench.ConvProbe.K_OF_KIND
    rec = {}
    order = []

    def sink(net, plan, rows, cm, esz, op_ms):
        probe = bench.ConvProbe()
        pair_cache = {}
        for i in range(plan["n_ops"]):
            op = plan["ops"][i]
            key = (op.map_kind, op.level_in, op.level_out)
            if key not in pair_cache:
                pair_cache[key] = probe._pairs(cm, op.map_kind, op.level_in, op.level_out, rows)
            pairs = pair_cache[key]
            cin, cout = plan["true_io"][i]
            k = K_OF_KIND[op.map_kind]
            v_in, v_out = rows[op.level_in], rows[op.level_out]
            nbytes = (v_in * cin + v_out * cout) * esz + k * cin * cout * esz + (8 * pairs if op.map_kind else 0)
            ident = (id(net), i)
            if ident not in rec:
                rec[ident] = dict(net=net.arch + "@%d" % v_out if i == 0 else "", i=i, kind=op.map_kind, lin=op.level_in,
                                  lout=op.level_out, v_in=v_in, v_out=v_out, cin=cin, cout=cout, k=k, pairs=pairs,
                                  bytes=nbytes, flops=2 * pairs * cin * cout, res=int(op.res_buf >= 0), us=[])
                order.append(ident)
            rec[ident]["us"].append(op_ms[i] * 1e3)

    for _ in range(3):
        bench.one_step(model, b, t)
    U.MinkUNet.OP_TIMING_SINK = sink
    for _ in range(reps):
        bench.one_step(model, b, t)
    U.MinkUNet.OP_TIMING_SINK = None
    torch.cuda.synchronize()
    print("%-18s %3s %4s %2s>%2s %7s %7s %4s %4s %3s %8s %9s %8s %7s %6s" % ("net", "op", "kind", "li", "lo", "v_in", "v_out", "cin",
                                                                       "cout", "K", "pairs", "MB", "GFLOP", "us", "frac"))
    tot_us = tot_b = 0
    lv = {}
    for ident in order:
        r = rec[ident]
        us = float(np.median(r["us"]))
        tot_us += us
        tot_b += r["bytes"]
        a = lv.setdefault(r["lout"], [0, 0.0, 0])
        a[0] += 1; a[1] += us; a[2] += r["bytes"]
        print("%-18s %3d %4d %2d>%2d %7d %7d %4d %4d %3d %8d %9.3f %8.3f %7.1f %6.3f" % (
            r["net"], r["i"], r["kind"], r["lin"], r["lout"], r["v_in"], r["v_out"], r["cin"], r["cout"], r["k"], r["pairs"],
            r["bytes"] / 1e6, r["flops"] / 1e9, us, r["bytes"] / (us * 1e-6) / 8e12))
    print("total: %d ops, %.1f us, %.1f MB, frac %.4f" % (len(order), tot_us, tot_b / 1e6, tot_b / (tot_us * 1e-6) / 8e12))
    for l in sorted(lv):
        c, us, by = lv[l]
        print("  out level %d: %3d ops %8.1f us (%.1f us/op) frac %.4f" % (l, c, us, us / c, by / (us * 1e-6) / 8e12))


if __name__ == "__main__":
    main()
