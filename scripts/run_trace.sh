#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --no-cpu-baseline --steps 80 --warmup 16 ${BENCH_ARGS} > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python scripts/analyze_trace.py $f
