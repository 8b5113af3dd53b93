#!/bin/bash
# A/B of one environment switch on a training / detection step, interleaved in one box: tools/ab_env.sh VAR "0 1" workload stage [reps] [steps]
VAR=$1; VALS=$2; WL=$3; STAGE=$4; REPS=${5:-4}; STEPS=${6:-12}
for r in $(seq $REPS); do for v in $VALS; do
  env $VAR=$v python bench.py --workload $WL --stage $STAGE --steps $STEPS --warmup 3 --no-cpu-baseline --no-extra-passes --no-side-configs --no-kernel-timing 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$VAR=$v $WL', d['ms_per_step'])"
done; done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3; if(!(k in m)||$3<m[k])m[k]=$3} END{for(k in a)print k, "min", m[k], "all", a[k]}'
