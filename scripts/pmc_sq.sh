#!/bin/bash
# SQ issue/wait counters of every kernel of a short bench run (one pass, 8 SQ counters).
# Usage (through gpurun, from the repo root): scripts/pmc_sq.sh tag ; then scripts/pmc_sq.py gpurun_out/<tag>_sq/s_results.db
set -e
tag=${1:-rXX}
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/${tag}_sq -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/${tag}_sq.log 2>&1
