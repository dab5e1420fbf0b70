#!/bin/bash
# kernel time per forward of the serving front: scripts/served_kstats.sh <tag> B F   -> gpurun_out/<tag>_kernel_stats.csv + summary
R=${GRAFT_REPO_ROOT:?run through gpurun}
TAG=$1; B=$2; F=$3
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python scripts/served_probe.py $B $F 2 > gpurun_out/${TAG}_kt.log 2>&1
cp $(find /tmp/kt_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/${TAG}_kt.log
python scripts/kstats_summary.py gpurun_out/${TAG}_kernel_stats.csv 30
