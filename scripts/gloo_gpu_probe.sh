#!/bin/bash
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2 PYTHONFAULTHANDLER=1
pids=()
for r in 0 1; do
  RANK=$r timeout -s ABRT -k 10 45 python scripts/gloo_gpu_probe.py > gpurun_out/gloo_probe_rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
echo "exit status $rc"; grep "rank" gpurun_out/gloo_probe_rank0.log | tail -3
exit $rc
