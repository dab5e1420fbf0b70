#!/bin/bash
# two rocprofv3 --pmc passes over the default bench command (separate runs, no trace domains), summarised into
# profiles/r01_pmc_summary.json by scripts/summarize_pmc.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}
# bench.py sets this with os.environ.setdefault, but under rocprofv3 the profiler has initialised the runtime before
# Python starts: export it in the shell so that the profiled run uses the same 8 hardware queues as the plain run
export GPU_MAX_HW_QUEUES=8
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $d -o bench -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 > /tmp/pmc_$c.log 2>&1
  tail -1 /tmp/pmc_$c.log | cut -c1-120
  f=$(find $d -name "bench_counter_collection.csv" | head -1)
  mkdir -p /tmp/pmc_flat_$c && cp $f /tmp/pmc_flat_$c/bench_counter_collection.csv
done
python scripts/summarize_pmc.py /tmp/pmc_flat_FETCH_SIZE /tmp/pmc_flat_WRITE_SIZE gpurun_out/r01_pmc_summary.json
