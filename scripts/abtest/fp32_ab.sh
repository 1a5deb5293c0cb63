#!/bin/bash
# alternating A/B of the fp32 step at the benched workload: ab/lib_base.so against the in-tree library
for i in 1 2; do
  for lib in ab/lib_base.so unet-based-network-traffic-prediction_amd/libunet_tm_hip.so; do
    timeout -k 10 200 python scripts/with_lib.py $lib bench.py --dtype fp32 --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$lib'[:12], 'ms_per_step', d['ms_per_step'], {n: round(1e3*v['ms_per_step'],1) for n,v in k.items() if 'wgrad' in n})" || exit 1
  done
done
