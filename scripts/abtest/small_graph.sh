#!/bin/bash
# the small training workload (16x16 images, batch 32) with and without HIP-graph replay, loop body and fused step
for i in 1 2; do
  for extra in "" "--fused-step"; do
    for g in 0 1; do
      UNET_TM_GRAPH=$g timeout -k 10 200 python bench.py --workload cfg1_abilene16_b32 $extra --no-cpu-baseline --steps 300 --warmup 30 --profile-steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
n=sum(v['launches_per_step'] for v in d.get('kernels', {}).values())
print('graph=$g', '$extra', 'ms_per_step', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'gpu kernel ms', d.get('gpu_kernel_ms_per_step'), 'launches', n)" || exit 1
    done
  done
done
