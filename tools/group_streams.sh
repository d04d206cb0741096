#!/bin/bash
# Same box, job level: windows per CNN pass ($PASSES) x analyzer streams ($STREAMS): rate, board power, clock.
for g in ${PASSES:-1024 512}; do
  for st in ${STREAMS:-2 3 4}; do
    timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events --streams $st --group-windows $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('power') or {}
print('[group=$g streams=$st]', round(d['value']/1e6,4), 'M windows/s', p.get('avg_W'), 'W', p.get('sclk_MHz_avg'), 'MHz')"
  done
done
