# in-flight throughput against the GIL switch interval of the bench's host threads (Python default 5 ms)
for sw in 0 0.001 0.0002 0.00005 0 0.0002 0.001 0.00005; do
  echo -n "switch=$sw: "
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 120 --switch-interval $sw 2>/dev/null | grep "^{" | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['timed_blocks']['value_p10'], b['timed_blocks']['value_p90'])"
done
