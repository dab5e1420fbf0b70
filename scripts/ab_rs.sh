#!/bin/bash
# A/B on one box: the row-stationary family off / on, alternating, 4 scenes in flight and 1
for r in 1 2; do
  for v in "PBN_CONV_RS=0" "PBN_CONV_RS=1"; do
    for f in 4 1; do
      line=$(env $v python bench.py --no-extras --inflight $f 2>/dev/null | grep "^{")
      echo "$v inflight=$f $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.1f ms %.3f" % (d["value"], d["ms_per_step"]), "stages", {k: round(v,3) for k,v in d.get("stages_ms",{}).items()} if f"{0}"=="x" else "")')"
    done
  done
done
