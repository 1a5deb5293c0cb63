#!/bin/bash
# Every bench workload once (through gpurun, from the repo root):
#   scripts/all_workloads.sh r03a
# keeps the FULL bench line of each run (roofline of its dominant kernel, per-kernel table) in
# gpurun_out/<tag>_workloads.jsonl and a one-line-per-run digest in gpurun_out/<tag>_workloads.txt
set -e
tag=${1:-rXX}
out=gpurun_out/${tag}_workloads.txt; jl=gpurun_out/${tag}_workloads.jsonl
: > $out; : > $jl
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 20 "$@" 2>/dev/null | tail -n 1 > gpurun_out/.wl_line.json
  cat gpurun_out/.wl_line.json >> $jl
  python -c "
import sys, json
d = json.loads(open('gpurun_out/.wl_line.json').read())
r = d.get('roofline') or {}
print('$*', d['config']['workload'], d['dtype'], d['ms_per_step'], round(d['value'], 1), d['unit'],
      '| dominant', r.get('kernel'), r.get('bound'), r.get('achieved'), r.get('unit'), 'frac', r.get('frac'))" >> $out; }
run --workload cfg1_abilene16_b32 --dtype fp32
run --workload cfg1_abilene16_b32
run --workload cfg1_abilene16_b32 --fused-step
run --workload cfg2_geant32_b512 --dtype fp32
run --workload cfg2_geant32_b512
run --workload cfg2_geant32_b512 --input-pipeline
run --workload cfg2_geant32_b512 --stock-optim
run --workload cfg4_deep128_b64
run --workload cfg5_infer64
run --workload cfg5_infer64 --batch 1
run --workload cfg5_infer64 --dtype fp32 --batch 1
rm -f gpurun_out/.wl_line.json
cat $out
