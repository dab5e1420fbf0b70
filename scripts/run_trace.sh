#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}
# bench.py sets this with os.environ.setdefault, but under rocprofv3 the profiler has initialised the runtime before
# Python starts: export it in the shell so that the profiled run uses the same 8 hardware queues as the plain run
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --no-cpu-baseline --steps 80 --warmup 16 ${BENCH_ARGS} > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python scripts/analyze_trace.py $f
