#!/bin/bash
# Development aid: step time with and without launch graphs at small batches.
for cfg in "cfg1_abilene16_b32 32" "cfg1_abilene16_b32 128" "cfg2_geant32_b512 64" "cfg2_geant32_b512 128"; do
  set -- $cfg
  for g in 0 1; do
    UNET_TM_GRAPH=$g timeout -k 5 100 python bench.py --workload $1 --batch $2 --steps 50 --warmup 10 --no-cpu-baseline --profile-steps 0 2>&1 \
      | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 batch $2 graph $g:', d['ms_per_step'], 'ms, host', d['host_enqueue_ms_per_step'])"
  done
done
