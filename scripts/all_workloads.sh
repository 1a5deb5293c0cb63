#!/bin/bash
# Every bench workload once (through gpurun, from the repo root): one line per run in gpurun_out/all_workloads.txt
set -e
out=gpurun_out/all_workloads.txt; : > $out
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 20 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['config']['workload'], d['dtype'], d['ms_per_step'], round(d['value'], 1), d['unit'])" >> $out; }
run --workload cfg1_abilene16_b32 --dtype fp32
run --workload cfg1_abilene16_b32
run --workload cfg2_geant32_b512 --dtype fp32
run --workload cfg2_geant32_b512
run --workload cfg2_geant32_b512 --input-pipeline
run --workload cfg2_geant32_b512 --stock-optim
run --workload cfg4_deep128_b64
run --workload cfg5_infer64
cat $out
