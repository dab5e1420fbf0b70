cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktp -- python scripts/probe_prepare.py > /dev/null 2>&1
f=$(find /tmp/ktp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
want = ("k_sort_pass", "k_pyramid", "k_maps_down", "k_maps_top", "k_insert_bbox", "k_unique_keys", "k_flag", "k_fill_ranges")
# take the LAST 600 matching rows (steady state) and group by (kernel, grid size)
agg = collections.OrderedDict()
for r in rows[-4000:]:
    n = r["Kernel_Name"]
    k = next((w for w in want if w in n), None)
    if not k: continue
    key = (k, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, []); a.append(d)
for (k, g), v in agg.items():
    print("%-16s grid %-8s calls %4d  avg %.1f us  min %.1f" % (k, g, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
PY
