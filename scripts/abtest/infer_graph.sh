#!/bin/bash
# batch-1..32 inference latency with and without HIP-graph replay of the plan's launch sequence
for b in 1 8 32; do
  for g in 0 1 0 1; do
    UNET_TM_GRAPH=$g timeout -k 10 200 python bench.py --workload cfg5_infer64 --batch $b --steps 300 --warmup 20 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('graph=$g batch=%5d  %9.4f ms/batch  host %7.4f ms' % (d['config']['per_gpu_batch'], d['ms_per_step'], d['host_enqueue_ms_per_step']), flush=True)" || exit 1
  done
done
