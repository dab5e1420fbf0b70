#!/bin/bash
# per-kernel table of one bench configuration: scripts/kstats.sh <tag> [bench args]   -> gpurun_out/<tag>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:?run through gpurun}
TAG=$1; shift
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python bench.py --no-extras "$@" > gpurun_out/${TAG}_kt.log 2>&1
cp $(find /tmp/kt_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/${TAG}_kt.log | cut -c1-300
