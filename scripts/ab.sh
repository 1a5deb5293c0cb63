#!/bin/bash
# In-call A/B of two builds of the library (ab/old.so, ab/new.so): alternates them under bench.py.
# Usage: scripts/ab.sh [rounds] [bench args...]; results in gpurun_out/ab_*.json
set -e
rounds=${1:-2}; shift || true
lib=unet-based-network-traffic-prediction_amd/libunet_tm_hip.so
for i in $(seq 1 $rounds); do
  for v in old new; do
    cp ab/$v.so $lib
    timeout -k 10 150 python bench.py --steps 200 --warmup 30 --no-cpu-baseline "$@" > gpurun_out/ab_${v}_$i.json 2>&1
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${v}_$i.json").read().strip().splitlines()[-1])
print("$v", $i, d["ms_per_step"], d["value"], flush=True)
PY
  done
done
cp ab/new.so $lib
