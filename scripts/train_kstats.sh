#!/bin/bash
# per-kernel table of the configs[2] training step: scripts/train_kstats.sh <tag>  -> gpurun_out/<tag>_train_kernel_stats.csv
R=${GRAFT_REPO_ROOT:?run through gpurun}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python scripts/train_step.py --steps 4 --warmup 2 "$@" > gpurun_out/${TAG}_train_kt.log 2>&1
cp $(find /tmp/kt_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_train_kernel_stats.csv
tail -1 gpurun_out/${TAG}_train_kt.log | cut -c1-300
python - "$TAG" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open('gpurun_out/%s_train_kernel_stats.csv' % sys.argv[1])))
steps = next(int(r['Calls']) for r in rows if 'k_centers' in r['Name'])
g = collections.Counter(); gc = collections.Counter()
keys = ['k_wgrad_ring', 'k_wgrad16', 'k_wgrad_reduce', 'k_wgrad<', 'k_bn_apply', 'k_bn_partial', 'k_bn_stats_final', 'k_bn_bwd_final', 'k_spconv_wave',
        'k_spconv<', 'k_spconv_reduce', 'k_pair', 'copyBuffer', 'fillBuffer', 'FusedAdam', 'at::native', 'rocprim', 'k_kernel_maps', 'k_pack', 'k_centers', 'k_count', 'k_union']
for r in rows:
    n = r['Name']
    k = next((k for k in keys if k in n), 'other')
    g[k] += int(r['TotalDurationNs']); gc[k] += int(r['Calls'])
tot = sum(g.values()); calls = sum(gc.values())
print("steps %d: %.2f ms of kernels per step, %d launches per step" % (steps, tot / 1e6 / steps, calls / steps))
for k, v in g.most_common():
    print("  %-18s %7.2f ms/step %6.0f calls/step  avg %6.1f us" % (k, v / 1e6 / steps, gc[k] / steps, v / gc[k] / 1e3))
PY
