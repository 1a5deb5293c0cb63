#!/bin/bash
# alternating comparison of the step time over several libraries: step_abc.sh N lib1 lib2 ...
n=$1; shift
for i in $(seq $n); do
  for lib in "$@"; do
    timeout -k 10 200 python scripts/with_lib.py $lib bench.py --no-cpu-baseline --steps 200 --warmup 30 --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib'.split('/')[-1][:22], 'ms_per_step', d['ms_per_step'])" || exit 1
  done
done
