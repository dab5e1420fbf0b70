#!/bin/bash
# the device front of the size-exact forward on / off by scene size: 1 and 3 copies of the configs[1] scene, the configs[3] scene
one() { python bench.py "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['timed_blocks']['cv'], d['config']['one_scene_in_flight_ms_per_scene'], d['config']['points_per_step'])
"; }
export PBNET_DEVICE_FRONT_MAX_POINTS=9999999
for c in 1 3; do for v in 0 1 0 1; do echo -n "copies $c device_front $v: "; PBNET_DEVICE_FRONT=$v one --copies $c --no-extras --steps 30; done; done
for v in 0 1 0 1; do echo -n "c4 device_front $v: "; PBNET_DEVICE_FRONT=$v one --workload c4 --no-extras --steps 16 --warmup 4; done
