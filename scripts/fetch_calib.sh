#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts for this path's access patterns (scripts/micro/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
R=${GRAFT_REPO_ROOT:?run through gpurun}
cd /tmp && export TMPDIR=/tmp
cd $R && mkdir -p gpurun_out
B=$R/scripts/micro/fetch_calib
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/micro/fetch_calib.hip -o $B
for rb in ${1:-192} ${2:-64}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fc_$c
    timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/fc_$c -o fc -- $B $rb > /tmp/fc_$c.log 2>&1
  done
  # gfx950's own request-size counters (counter_defs.yaml: TCC events 43-45): 32 / 64 / 128-byte read requests to the fabric
  rm -rf /tmp/fc_REQ
  timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d /tmp/fc_REQ -o fc -- $B $rb > /tmp/fc_REQ.log 2>&1
  python3 $R/scripts/fetch_calib_join.py $rb
done | tee gpurun_out/fetch_calib.txt
