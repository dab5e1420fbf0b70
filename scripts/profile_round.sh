#!/bin/bash
# Everything profiles/ holds for one milestone, from one box:  bash scripts/profile_round.sh r02_a
#   <tag>_bench.json                   the default bench.py line
#   <tag>_bench_kernel_stats.csv       rocprofv3 --kernel-trace --stats of the same command (no CPU baseline)
#   <tag>_inflight1_kernel_stats.csv   the same with one scene in flight
#   <tag>_{bench,inflight1,planned,graph}_concurrency.json   union of kernel intervals / wall of those traces (+ --forward-mode planned / graph)
#   <tag>_pmc_summary.json             HBM traffic per kernel family: two --pmc passes (FETCH_SIZE, WRITE_SIZE), own runs
#   <tag>_sq_conv_summary.json         SQ counters (MFMA busy, waits) of the convolution kernels, own pass
#   <tag>_train_step.json / <tag>_train_kernel_stats.csv   scripts/train_step.py (configs[2] on one rank) and its kernel table
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}
TAG=${1:?usage: profile_round.sh <tag>}
# bench.py sets this with os.environ.setdefault, but under rocprofv3 the profiler has initialised the runtime before Python
# starts: export it in the shell so that the profiled runs use the same 8 hardware queues as the plain run
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
cd $R
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 600 python bench.py 2> $O/bench.err | grep "^{" > $O/${TAG}_bench.json; tail -1 $O/${TAG}_bench.json | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --no-extras --min-seconds 3 > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python bench.py --no-extras --inflight 1 --min-seconds 3 > $O/kt1.log 2>&1
PBN_TRACE_JSON=$O/${TAG}_bench_concurrency.json python scripts/analyze_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/${TAG}_bench_concurrency.txt 2>&1; head -3 $O/${TAG}_bench_concurrency.txt
PBN_TRACE_JSON=$O/${TAG}_inflight1_concurrency.json python scripts/analyze_trace.py $(find $O/kt1 -name "*kernel_trace.csv" | head -1) > $O/${TAG}_inflight1_concurrency.txt 2>&1; head -3 $O/${TAG}_inflight1_concurrency.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
# (3 s of timed blocks: analyze_trace.py's window, 35-80 % of the trace, then lies inside the timed region)
# the sync-free forward as eager launches and from HIP graphs, four in flight: how much of the wall has a kernel running
for m in planned graph; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -- python bench.py --no-extras --forward-mode $m --min-seconds 3 > $O/kt_$m.log 2>&1
  PBN_TRACE_JSON=$O/${TAG}_${m}_concurrency.json python scripts/analyze_trace.py $(find $O/kt_$m -name "*kernel_trace.csv" | head -1) > $O/${TAG}_${m}_concurrency.txt 2>&1; head -3 $O/${TAG}_${m}_concurrency.txt
  grep "^{" $O/kt_$m.log | tail -1 | cut -c1-120
  rm -rf $O/kt_$m
done
cp $(find $O/kt1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_inflight1_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $d -o bench -- python bench.py --no-extras --steps 5 --warmup 2 --repeats 1 --min-seconds 0 > $O/pmc_$c.log 2>&1
  mkdir -p /tmp/pmc_flat_$c && cp $(find $d -name "bench_counter_collection.csv" | head -1) /tmp/pmc_flat_$c/bench_counter_collection.csv
done
python scripts/summarize_pmc.py /tmp/pmc_flat_FETCH_SIZE /tmp/pmc_flat_WRITE_SIZE $O/${TAG}_pmc_summary.json > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_sq -o sq -- python bench.py --no-extras --inflight 1 --steps 5 --warmup 2 --repeats 1 --min-seconds 0 > $O/pmc_sq.log 2>&1
python scripts/summarize_sq.py $(find /tmp/pmc_sq -name "sq_counter_collection.csv" | head -1) $(find /tmp/pmc_sq -name "sq_kernel_trace.csv" | head -1) $O/${TAG}_sq_conv_summary.json > $O/sq_summary.log 2>&1; tail -12 $O/sq_summary.log
# the training step of configs[2]: its line and its kernel table
timeout 300 python scripts/train_step.py --steps 8 --warmup 2 --phases 2> $O/train_step.err | grep "^{" > $O/${TAG}_train_step.json; tail -1 $O/${TAG}_train_step.json | cut -c1-200; grep phases $O/train_step.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -- python scripts/train_step.py --steps 5 --warmup 1 > $O/ktt.log 2>&1
cp $(find $O/ktt -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_kernel_stats.csv; rm -rf $O/ktt
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; rm -rf $O/kt $O/kt1
ls $O
