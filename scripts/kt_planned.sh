cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktp -- python scripts/host_profile_planned.py 100 > /tmp/ktp.log 2>&1
f=$(find /tmp/ktp -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
n=205.0   # forwards in the script: 5 warm + 100 timed + 100 profiled
def fam(n):
    for k,v in (("k_spconv_rsh","rsh"),("k_spconv_rs<","rs"),("k_spconv_wave","wave"),("k_spconv<","tile")):
        if k in n: return v
    for k in ("k_count","k_union","k_centers","k_border","k_noise","k_cell","k_member","k_relabel","k_tag","k_flatten","k_compress","k_sizes","k_keep","k_seg_off","k_cluster_num","k_compact_noise","k_copy_i32","k_hp"):
        if k in n: return "grouping"
    for k in ("k_insert","k_flag","k_unique","k_sort","k_pyramid","k_maps","k_kernel_map","k_stride","k_morton","k_apply_perm","k_fill","k_scan"):
        if k in n: return "coords"
    return "glue"
agg={}
for r in rows:
    a=agg.setdefault(fam(r["Name"]),[0,0.0]); a[0]+=int(r["Calls"]); a[1]+=float(r["TotalDurationNs"])/1e6
print({k:(round(v[0]/n,1), round(v[1]/n,3)) for k,v in sorted(agg.items())}, "total ms/forward", round(sum(v[1] for v in agg.values())/n,3))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:14]:
    print("   %-60s calls/fwd %5.1f avg %.1f us  ms/fwd %.3f" % (r["Name"][28:88], int(r["Calls"])/n, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/n))
PY
