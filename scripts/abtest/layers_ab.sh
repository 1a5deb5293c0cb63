#!/bin/bash
# A/B of the per-launch table and the step time: ab/lib_base.so against the in-tree library
# usage: ab/layers_ab.sh "<grep pattern for rows>"
pat=${1:-ups}
for lib in ab/lib_base.so unet-based-network-traffic-prediction_amd/libunet_tm_hip.so; do
  echo "== $lib"
  timeout -k 10 200 python scripts/with_lib.py $lib bench.py --no-cpu-baseline --steps 100 --warmup 20 --layers-out gpurun_out/.layers.txt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'gpu_kernel_ms', d.get('gpu_kernel_ms_per_step'))" || exit 1
  grep -E "$pat" gpurun_out/.layers.txt
done
