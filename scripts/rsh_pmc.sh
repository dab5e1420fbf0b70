#!/bin/bash
# SQ counters of the staged row-stationary kernel on one layer (two passes of 8 counters), printed per kernel name
R=${GRAFT_REPO_ROOT:?run through gpurun}
cd /tmp && export TMPDIR=/tmp
cd $R
export PBN_PROBE_CFGS=${PBN_PROBE_CFGS:-11000} PBN_PROBE_CASES=${PBN_PROBE_CASES:-"0,96,96"}
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/sq_counters.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d /tmp/rshpmc$i -o p -- python3 scripts/probe_rs.py > gpurun_out/rsh_pmc_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/rshpmc$i/**/p_counter_collection.csv',recursive=True)
if not f: print('no csv'); raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name'] in ('SQ_WAVE_CYCLES','SQ_INSTS_LDS'): cnt[k]+=1
for k,v in agg.items():
    if 'spconv' in k: print(k, 'launches',cnt[k], {c:round(x/max(cnt[k],1)) for c,x in v.items()})
PY
done
