"""Per-forward summary of a rocprofv3 kernel_stats.csv: launches and microseconds per kernel family (one forward = one k_centers call)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
fw = next(int(r["Calls"]) for r in rows if "k_centers" in r["Name"])
fam = {}
for r in rows:
    n = r["Name"]
    m = re.search(r"(k_\w+|rocprim|at::native::\w+|__amd_\w+)", n)
    key = m.group(1) if m else n[:40]
    if "k_spconv_wave" in n: key = "k_spconv_wave"
    elif "k_spconv_reduce" in n: key = "k_spconv_reduce"
    elif "k_spconv" in n: key = "k_spconv (tile)"
    f = fam.setdefault(key, [0, 0]); f[0] += int(r["Calls"]); f[1] += int(r["TotalDurationNs"])
tot_c = sum(v[0] for v in fam.values()) / fw; tot_t = sum(v[1] for v in fam.values()) / fw / 1e3
print("forwards %d; per forward: %.1f launches, %.1f us of kernels" % (fw, tot_c, tot_t))
groups = {"conv": ("k_spconv",), "coords": ("k_insert", "k_flag", "k_unique", "k_sort", "k_pyramid", "k_maps", "k_kernel_map", "k_stride", "k_morton", "k_apply_perm", "rocprim", "k_fill"),
          "grouping": ("k_centers", "k_count", "k_union", "k_border", "k_noise", "k_cell", "k_tag", "k_relabel", "k_compress", "k_flatten", "k_compact", "k_scan", "k_sizes", "k_members", "k_seg")}
gs = {g: [0, 0] for g in groups}; gs["other"] = [0, 0]
for k, v in fam.items():
    g = next((g for g, pre in groups.items() if any(k.startswith(p) for p in pre)), "other")
    gs[g][0] += v[0]; gs[g][1] += v[1]
for g, v in gs.items():
    print("  %-9s %6.1f launches %8.1f us" % (g, v[0] / fw, v[1] / fw / 1e3))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("    %-44s %6.1f x %7.1f us = %8.1f us" % (k, v[0] / fw, v[1] / v[0] / 1e3, v[1] / fw / 1e3))
