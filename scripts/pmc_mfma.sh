#!/bin/bash
# MFMA-busy counters of every kernel of a short bench run (one pass: 4 SQ counters + 1 GRBM).
# Usage (through gpurun, from the repo root): scripts/pmc_mfma.sh tag [extra bench.py arguments, e.g. --dtype fp32]
#   then, locally: scripts/pmc_sq.py gpurun_out/<tag>_sq/s_results.db gpurun_out/<tag>_mfma/m_results.db
set -e
tag=${1:-rXX}
shift || true
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $R/gpurun_out/${tag}_mfma -o m -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 "$@" > $R/gpurun_out/${tag}_mfma.log 2>&1
