#!/bin/bash
# same-box A/B of an environment switch on the hot path: tools/ab.sh VAR A_VALUE B_VALUE [bench args...]
V=$1; A=$2; B=$3; shift 3
for rep in 1 2 3; do
  for val in $A $B; do
    r=$(env $V=$val python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-passes --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])")
    echo "$V=$val ms_per_step $r"
  done
done
