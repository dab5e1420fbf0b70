#!/bin/bash
# Is the in-flight rate bound by the interpreter lock?  P processes x F scenes in flight each, started together on ONE GPU; the
# sum of their rates against one process with P x F in flight.  usage: probe_multiproc.sh "P:F P:F ..."
for pf in ${1:-"1:4 2:2 4:1 4:2"}; do
  P=${pf%%:*}; F=${pf##*:}
  rm -f /tmp/mp_*.json
  for i in $(seq 1 $P); do
    python bench.py --no-extras --inflight $F --min-seconds 6 --steps 60 2>/dev/null | grep "^{" > /tmp/mp_$i.json &
  done
  wait
  python - <<PY
import json,glob
v=[json.load(open(f))["value"] for f in sorted(glob.glob("/tmp/mp_*.json"))]
print("$P processes x $F in flight: per process", [round(x,1) for x in v], "sum %.1f scenes/s" % sum(v))
PY
done
