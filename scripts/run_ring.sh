#!/bin/bash
timeout 600 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu 2>&1 | tail -3
for ring in 2 3; do for sp in 768 384 256; do
  echo "== ring $ring split $sp"
  PBN_CONV_RING=$ring PBN_CONV_SPLIT=$sp PBN_PROBE_RWS=${RWS:-16,32} timeout 300 python scripts/probe_conv_ablate.py 2>&1 | grep dbg= | cut -c1-72
done; done
