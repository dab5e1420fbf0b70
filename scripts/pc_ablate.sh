#!/bin/bash
# Ablation of the pair-compacted kernel (PBN_CONV_DBG bits: 2 no MFMAs, 4 gathers out of range, 8 weight loads out of range, 16 no main
# loop, 32 no epilogue, 64 no write-back) on one layer of the bench scene: microseconds per launch from a HIP graph.
# usage: pc_ablate.sh "level,cin,cout[,k]" [cfg=13000]
export PBNET_HIP_LIB=$(dirname $0)/../pbnet_amd/libpbnet_hip_exp.so      # (make -C pbnet_amd/csrc experiments)
CASE=${1:-0,96,96}
CFG=${2:-13000}
for d in 0 2 4 8 64 12 78 16 48; do
  echo -n "dbg=$d: "
  PBN_CONV_DBG=$d PBN_PROBE_CFGS=$CFG PBN_PROBE_CASES="$CASE" python scripts/probe_rs.py 2>&1 | grep "rows="
done
