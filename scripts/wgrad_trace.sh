#!/bin/bash
# per-kernel durations of scripts/probe_wgrad.py (one layer shape per line of its output): gpurun_out/wgrad_trace.txt
R=${GRAFT_REPO_ROOT:?run through gpurun}
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_wg -- python scripts/probe_wgrad.py > gpurun_out/wgrad_probe.log 2>&1
python - <<'PY' > gpurun_out/wgrad_trace.txt
import csv, glob, collections
f = glob.glob('/tmp/kt_wg/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# consecutive runs of the same (kernel, grid)
runs = []
for r in rows:
    n = r['Kernel_Name']
    if 'wgrad' not in n: continue
    key = (n[:90], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('LDS_Block_Size'))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if runs and runs[-1][0] == key: runs[-1][1].append(d)
    else: runs.append([key, [d]])
agg = collections.OrderedDict()
for k, ds in runs:
    agg.setdefault(k, []).extend(ds)
for k, ds in agg.items():
    ds.sort()
    print("%-92s grid %-9s lds %-6s calls %4d  median %8.1f us  min %8.1f" % (k[0], k[1], k[2], len(ds), ds[len(ds)//2] / 1e3, ds[0] / 1e3))
PY
tail -8 gpurun_out/wgrad_probe.log
cat gpurun_out/wgrad_trace.txt
