#!/bin/bash
# r01_f: default bench (scenes in flight) + rocprofv3 kernel stats of the same command + one-scene-in-flight stats
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}
# bench.py sets this with os.environ.setdefault, but under rocprofv3 the profiler has initialised the runtime before
# Python starts: export it in the shell so that the profiled run uses the same 8 hardware queues as the plain run
export GPU_MAX_HW_QUEUES=8
O=$R/gpurun_out/prof_f
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --no-cpu-baseline > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python bench.py --no-cpu-baseline --inflight 1 > $O/kt1.log 2>&1
find $O -name "*kernel_stats.csv" | head
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
