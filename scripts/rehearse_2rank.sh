#!/bin/bash
# Two data-parallel ranks on ONE GPU (gloo stages through host memory; RCCL refuses two ranks per
# device): a rehearsal of bench.py's N > 1 flow on a one-GPU box.  Each rank is its own process with
# its own time limit; SIGABRT at the limit makes Python dump every thread's stack.
# Usage (through gpurun, from the repo root): scripts/rehearse_2rank.sh [bench args...]
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2
export UNET_TM_DIST_BACKEND=gloo UNET_TM_FORCE_DEVICE=0 PYTHONFAULTHANDLER=1
pids=()
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 45 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 "$@" \
    > gpurun_out/rehearse_rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
echo "exit status $rc"
tail -c 600 gpurun_out/rehearse_rank0.log
exit $rc
