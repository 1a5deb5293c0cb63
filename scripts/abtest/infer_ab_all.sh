#!/bin/bash
# A/B of the 64x64 inference workload over batch sizes: ab/lib_base.so against the in-tree library
for b in 2 8 16 32 64 128; do
  for lib in ab/lib_base.so unet-based-network-traffic-prediction_amd/libunet_tm_hip.so; do
    timeout -k 10 200 python scripts/with_lib.py $lib bench.py --workload cfg5_infer64 ${DT:+--dtype $DT} --batch $b --steps 300 --warmup 20 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib'[:12], d['dtype'], 'batch=%5d  %9.4f ms/batch  host %7.4f ms' % (d['config']['per_gpu_batch'], d['ms_per_step'], d['host_enqueue_ms_per_step']), flush=True)" || exit 1
  done
done
