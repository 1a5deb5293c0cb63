#!/bin/bash
# per-kernel table of the small training workload (16x16 images, batch 32, fused step); $1 = library (optional)
lib=${1:-unet-based-network-traffic-prediction_amd/libunet_tm_hip.so}
timeout -k 10 200 python scripts/with_lib.py $lib bench.py --workload cfg1_abilene16_b32 --fused-step --no-cpu-baseline --steps 300 --warmup 30 --profile-steps 20 --layers-out gpurun_out/small_layers.txt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
n=sum(v['launches_per_step'] for v in d['kernels'].values())
print('$lib: %.4f ms/step, host %.4f, %d launches, gpu kernel ms %.4f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], n, d['gpu_kernel_ms_per_step']))
for k,v in d['kernels'].items(): print('  %-22s x%-3d %6.1f us each %7.1f us' % (k, v['launches_per_step'], 1e3*v['ms_per_step']/v['launches_per_step'], 1e3*v['ms_per_step']))
"
cat gpurun_out/small_layers.txt
