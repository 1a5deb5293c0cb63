#!/bin/bash
# In-call comparison of several builds of the library (ab/<name>.so): alternates them under bench.py
# and writes each build's per-layer table.  Usage: scripts/ab3.sh rounds name1 name2 ...
set -e
rounds=$1; shift
lib=unet-based-network-traffic-prediction_amd/libunet_tm_hip.so
cp $lib /tmp/keep.so
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    cp ab/$v.so $lib
    timeout -k 10 150 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --layers-out gpurun_out/ab_${v}_layers.txt > gpurun_out/ab_${v}_$i.json 2>&1
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${v}_$i.json").read().strip().splitlines()[-1])
print("$v", $i, d["ms_per_step"], d["value"], flush=True)
PY
  done
done
cp /tmp/keep.so $lib
