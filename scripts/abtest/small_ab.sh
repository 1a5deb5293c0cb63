#!/bin/bash
# alternating A/B of the small training workload (16x16 images, batch 32), loop body and fused step:
# ab/lib_base.so against the in-tree library
for i in 1 2; do
  for extra in "" "--fused-step"; do
    for lib in ab/lib_base.so unet-based-network-traffic-prediction_amd/libunet_tm_hip.so; do
      timeout -k 10 200 python scripts/with_lib.py $lib bench.py --workload cfg1_abilene16_b32 $extra --no-cpu-baseline --steps 300 --warmup 30 --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib'[:12], '$extra', 'ms_per_step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])" || exit 1
    done
  done
done
