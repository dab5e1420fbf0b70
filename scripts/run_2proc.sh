#!/bin/bash
run() { timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 30 --inflight $1 2>&1 | tail -1 | cut -c60-100; }
echo "== 2 processes x 2 in flight"; run 2 > /tmp/a.log & run 2 > /tmp/b.log; wait; cat /tmp/a.log /tmp/b.log
echo "== 2 processes x 3 in flight"; run 3 > /tmp/a.log & run 3 > /tmp/b.log; wait; cat /tmp/a.log /tmp/b.log
echo "== 2 processes x 4 in flight"; run 4 > /tmp/a.log & run 4 > /tmp/b.log; wait; cat /tmp/a.log /tmp/b.log
echo "== 3 processes x 2 in flight"; run 2 > /tmp/a.log & run 2 > /tmp/c.log & run 2 > /tmp/b.log; wait; cat /tmp/a.log /tmp/b.log /tmp/c.log
echo "== 4 processes x 1 in flight"; run 1 > /tmp/a.log & run 1 > /tmp/c.log & run 1 > /tmp/d.log & run 1 > /tmp/b.log; wait; cat /tmp/a.log /tmp/b.log /tmp/c.log /tmp/d.log
