#!/bin/bash
# Inference throughput / latency sweep of the 64x64 workload over batch sizes (run on the GPU box):
#   scripts/infer_sweep.sh > gpurun_out/infer_sweep.txt
set -e
for g in 0 1; do
  for b in 1 2 8 32 128 512 2048 8192; do
    steps=200; [ $b -ge 2048 ] && steps=30
    UNET_TM_GRAPH=$g timeout -k 10 200 python bench.py --workload cfg5_infer64 --batch $b --steps $steps --warmup 10 --no-cpu-baseline --profile-steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
n=sum(v['launches_per_step'] for v in d.get('kernels', {}).values())
print('graph=$g batch=%5d  %9.4f ms/batch  %10.1f TMs/s  host %7.4f ms  %3d launches per forward' % (d['config']['per_gpu_batch'], d['ms_per_step'], d['value'], d['host_enqueue_ms_per_step'], n), flush=True)"
  done
done
