#!/bin/bash
# Collects a round's measurement artifacts on the GPU box (run through gpurun from the repo root):
#   scripts/round_profile.sh r01c
# writes gpurun_out/<tag>_*: the default bench line, the same command under rocprofv3 --kernel-trace
# --stats, the per-layer table, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) of a short run.
# Afterwards, locally:  scripts/prof_summary.py / scripts/pmc_traffic.py turn the databases into profiles/.
set -e
tag=${1:-rXX}
R=$PWD
out=$R/gpurun_out
timeout -k 10 300 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --layers-out $out/${tag}_layers.txt > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o k -- python3 $R/bench.py > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/${tag}_pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/${tag}_pmc_write.log 2>&1
cd $R
ls -la $out/${tag}_prof $out/${tag}_pmc_fetch $out/${tag}_pmc_write
