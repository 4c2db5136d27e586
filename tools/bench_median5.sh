#!/bin/bash
# SURVEY 8(d) timing protocol: 20 warm-up + 100 timed steps, median of 5 repeats (one bench.py process per repeat)
for i in 1 2 3 4 5; do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null
done | python -c "
import json, sys
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
ms = sorted(r['ms_per_step'] for r in rows)
print(json.dumps({'protocol': '20 warm-up + 100 timed steps, 5 repeats', 'ms_per_step_all': ms, 'ms_per_step_median': ms[len(ms)//2],
                  'images_per_sec_median': round(256e3 / ms[len(ms)//2], 1), 'launch_modes': sorted(set(r['launch'] for r in rows))}))"
