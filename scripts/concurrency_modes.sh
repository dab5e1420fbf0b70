#!/bin/bash
# gpu_active_frac of the four timed modes (size-exact 4 in flight / 1 in flight, planned eager, graph replay): kernel traces of bench.py
# with 3 s of timed blocks each -> gpurun_out/<tag>/<tag>_{bench,inflight1,planned,graph}_concurrency.{json,txt}   usage: concurrency_modes.sh <tag>
R=${GRAFT_REPO_ROOT:?run through gpurun}
TAG=${1:?tag}
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
cd $R
O=$R/gpurun_out/$TAG; mkdir -p $O
run() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/cm_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/cm_$name -- python bench.py --no-extras --min-seconds 3 "$@" > $O/cm_$name.log 2>&1
  PBN_TRACE_JSON=$O/${TAG}_${name}_concurrency.json python scripts/analyze_trace.py $(find /tmp/cm_$name -name "*kernel_trace.csv" | head -1) > $O/${TAG}_${name}_concurrency.txt 2>&1
  echo "$name: $(grep '^{' $O/cm_$name.log | tail -1 | cut -c1-90) $(head -2 $O/${TAG}_${name}_concurrency.txt | tail -1)"
  rm -rf /tmp/cm_$name
}
run bench
run inflight1 --inflight 1
run planned --forward-mode planned
run graph --forward-mode graph
