#!/bin/bash
# same-box A/B of one environment switch on a bench.py command line: ab_env.sh VAR "<bench args>" [rounds]
VAR=$1; ARGS=$2; R=${3:-3}
for i in $(seq 1 $R); do for v in 1 0; do
  env $VAR=$v timeout 300 python bench.py $ARGS --no-cpu-baseline --no-extra-passes --no-side-configs --no-kernel-timing 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$VAR=$v', d['ms_per_step'])"
done; done
