#!/bin/bash
# HBM traffic per convolution op (scripts/pmc_by_op.py): two --pmc passes with one scene in flight + the op table.  -> gpurun_out/pmc_by_op.txt
R=${GRAFT_REPO_ROOT:?run through gpurun}
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmcop_$c; rm -rf $d /tmp/pmcop_flat_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $d -o bench -- python bench.py --no-extras --inflight 1 --steps 4 --warmup 1 --repeats 1 --min-seconds 0 > gpurun_out/pmcop_$c.log 2>&1
  mkdir -p /tmp/pmcop_flat_$c && cp $(find $d -name "bench_counter_collection.csv" | head -1) /tmp/pmcop_flat_$c/bench_counter_collection.csv
done
python scripts/op_table.py 2>/dev/null > gpurun_out/pmcop_op_table.txt
python scripts/pmc_by_op.py /tmp/pmcop_flat_FETCH_SIZE /tmp/pmcop_flat_WRITE_SIZE gpurun_out/pmcop_op_table.txt > gpurun_out/pmc_by_op.txt 2>&1
tail -12 gpurun_out/pmc_by_op.txt
