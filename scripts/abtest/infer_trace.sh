#!/bin/bash
# kernel trace of the 64x64 inference workload at batch $1: start/end of every launch of a few forward passes
R=$PWD; out=$R/gpurun_out; b=${1:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf $out/infer_trace_b$b
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/infer_trace_b$b -o k -- python3 $R/bench.py --workload cfg5_infer64 --batch $b --steps 100 --warmup 20 --no-cpu-baseline --profile-steps 0 > $out/infer_trace_b$b.json 2> $out/infer_trace_b$b.err
cd $R
python - <<PY
import csv, glob
f = glob.glob("$out/infer_trace_b$b/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-23*40:]      # the last forty forward passes
import collections
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    n = b["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    dur[n].append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
    gap[n].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
tot = 0
for n in dur:
    d = sorted(dur[n]); g = sorted(gap[n])
    print("%-62s x%-4d dur med %6.2f us  gap-before med %6.2f us" % (n, len(d), d[len(d)//2]/1e3, g[len(g)//2]/1e3))
PY
