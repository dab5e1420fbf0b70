for cfg in "30000 10" "20000 10" "60000 10" "30000 5" "30000 20" "60000 20" "30000 10"; do
  set -- $cfg
  echo -n "MAX_ROWS=$1 GMACS=$2: "
  PBN_WAVE_MAX_ROWS=$1 PBN_WAVE_MAX_GMACS=$2 timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 120 2>/dev/null | grep "^{" | python -c "import sys,json; b=json.loads(sys.stdin.read()); r=b['roofline']; print(b['value'], b['timed_blocks']['value_p10'], b['timed_blocks']['value_p90'], 'inflight us', r['avg_launch_us'], 'alone us', r['one_scene_in_flight']['avg_launch_us'])"
done
