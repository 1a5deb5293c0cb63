#!/bin/bash
# per-kernel table (launches, us) of the 64x64 inference workload at one batch size: $1 = batch, $2 = library (optional)
lib=${2:-unet-based-network-traffic-prediction_amd/libunet_tm_hip.so}
timeout -k 10 200 python scripts/with_lib.py $lib bench.py --workload cfg5_infer64 ${DT:+--dtype $DT} --batch $1 --steps 300 --warmup 20 --no-cpu-baseline --profile-steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
n=sum(v['launches_per_step'] for v in d['kernels'].values())
print('$lib batch %d: %.4f ms/batch, %d launches per forward' % (d['config']['per_gpu_batch'], d['ms_per_step'], n))
for k,v in d['kernels'].items(): print('  %-18s x%-3d %6.1f us each' % (k, v['launches_per_step'], 1e3*v['ms_per_step']/v['launches_per_step']))
"
