#!/bin/bash
# A/B of the convolution kernel families on one box: tests, per-op table and short bench runs per PBN_CONV_FAMILY
# (0 = workgroup-tile LDS-ring kernels only, 1 = wave-autonomous kernels everywhere, 2 = by level size, the default)
R=${GRAFT_REPO_ROOT:?run through gpurun}
cd $R
O=gpurun_out/${1:-ab}
mkdir -p $O
python -m pytest tests/test_backbone_gpu.py -m gpu -q -x -s > $O/test_backbone.log 2>&1; echo "backbone tests rc=$?"; tail -2 $O/test_backbone.log
PBN_CONV_FAMILY=1 python -m pytest tests/test_backbone_gpu.py tests/test_pbnet_gpu.py tests/test_bench_workload_gpu.py -m gpu -q -x > $O/test_family1.log 2>&1; echo "family1 tests rc=$?"; tail -2 $O/test_family1.log
python -m pytest tests/test_train_gpu.py -m gpu -q -x -s > $O/test_train.log 2>&1; echo "train tests rc=$?"; tail -2 $O/test_train.log
for fam in ${FAMS:-0 2 1}; do
  PBN_CONV_FAMILY=$fam python scripts/probe_ops.py > $O/probe_ops_fam$fam.log 2>&1
  grep "total" $O/probe_ops_fam$fam.log | tr '\n' ' '; echo " <- family $fam"
  PBN_CONV_FAMILY=$fam python bench.py --no-extras --steps 40 --repeats 3 > $O/bench_fam$fam.json 2>$O/bench_fam$fam.err
  python - <<PY
import json
d=json.loads(open("$O/bench_fam$fam.json").read().strip().splitlines()[-1])
print("family $fam: %.1f scenes/s, one-in-flight %.3f ms, roofline frac %.4f (alone %.4f)" % (d["value"], d["config"]["one_scene_in_flight_ms_per_scene"], d["roofline"]["frac"], d["roofline"]["one_scene_in_flight"]["frac"]))
PY
done
