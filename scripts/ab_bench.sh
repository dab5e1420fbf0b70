#!/bin/bash
# alternating A/B of two library builds on the bench line: scripts/ab_bench.sh "prev new" [rounds] [inflight]
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=$1; ROUNDS=${2:-3}; INF=${3:-4}
for r in $(seq 1 $ROUNDS); do
  for l in $LIBS; do
    f=$R/pbnet_amd/libpbnet_hip_$l.so; [ "$l" = new ] && f=$R/pbnet_amd/libpbnet_hip.so
    echo "$l inflight $INF: $(PBNET_HIP_LIB=$f python bench.py --no-extras --inflight $INF --repeats 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  done
done
