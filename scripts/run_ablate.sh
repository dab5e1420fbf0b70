#!/bin/bash
# PBN_CONV_DBG ablation sweep
export PBN_PROBE_CASES=${CASES:-0,1}
for d in ${DBGS:-0 1 2 4 8 12 14 16}; do
  PBN_CONV_DBG=$d PBN_PROBE_RWS=${RWS:-32} timeout 300 python scripts/probe_conv_ablate.py 2>&1 | grep dbg= | cut -c1-72
done
