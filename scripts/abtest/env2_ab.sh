#!/bin/bash
# alternating A/B of the step time between two environment settings: ab/env2_ab.sh VAR=a VAR=b [rounds] [extra bench args]
a=$1; b=$2; n=${3:-3}; shift 3
for i in $(seq $n); do
  for kv in $a $b; do
    env $kv timeout -k 10 200 python bench.py --no-cpu-baseline --steps 200 --warmup 30 --profile-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$kv', 'ms_per_step', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'))" || exit 1
  done
done
