#!/bin/bash
# Board power and shader clock of the timed region by analyzer streams (bench.py's own `power` object), default launch set and $1.
for st in 1 2 3 4; do
  for arm in new old; do
    if [ $arm = old ]; then v="--sep-variant $1"; else v=""; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events --streams $st $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('power') or {}
print('[$arm streams=$st]', round(d['value']/1e6,4), 'M windows/s', p.get('avg_W'), 'W', p.get('sclk_MHz_avg'), 'MHz')"
  done
done
