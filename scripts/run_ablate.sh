#!/bin/bash
# PBN_CONV_DBG ablation sweep for the two big stride-1 levels
export PBN_PROBE_CASES=0,1
PBN_PROBE_RWS=16,32,64 timeout 300 python scripts/probe_conv_ablate.py 2>&1 | grep dbg=
for d in 1 2 4 8 12 14 16; do
  PBN_CONV_DBG=$d PBN_PROBE_RWS=32 timeout 300 python scripts/probe_conv_ablate.py 2>&1 | grep dbg=
done
