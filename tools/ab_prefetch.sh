#!/bin/bash
# same-box A / B of the frame-head prefetch variants (bench.py, headline workload, no extra passes)
B="python bench.py --no-cpu-baseline --no-extra-passes --no-side-configs --no-kernel-timing --steps 40 --warmup 8"
for rep in 1 2; do
  for v in "noprefetch:--no-prefetch" "head_only:" "head+img:"; do
    name=${v%%:*}; flag=${v#*:}
    if [ "$name" = "head+img" ]; then export DF3D_IMGPROJ_AHEAD=1; else export DF3D_IMGPROJ_AHEAD=0; fi
    $B $flag 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', d['ms_per_step'], d['value'])"
  done
done
