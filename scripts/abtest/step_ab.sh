#!/bin/bash
# alternating A/B of the step time: ab/lib_base.so against the in-tree library, N rounds
n=${1:-3}
for i in $(seq $n); do
  for lib in ab/lib_base.so unet-based-network-traffic-prediction_amd/libunet_tm_hip.so; do
    timeout -k 10 200 python scripts/with_lib.py $lib bench.py --no-cpu-baseline --steps 200 --warmup 30 --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib'[:12], 'ms_per_step', d['ms_per_step'])" || exit 1
  done
done
