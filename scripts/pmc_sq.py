"""Per-kernel SQ issue/wait breakdown from a rocprofv3 --pmc database written by scripts/pmc_sq.sh.
Columns are shares of SQ_WAVE_CYCLES (summed over the waves of a launch, averaged over launches)."""
import re
import sqlite3
import sys
from collections import defaultdict

from prof_summary import short

db = sqlite3.connect(sys.argv[1])
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for name, counter, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
    acc[short(name)][counter] += value
for name, in db.execute("select kernel_name from counters_collection where counter_name='SQ_WAVE_CYCLES'"):
    n[short(name)] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])
print(f"{'kernel':44s} {'launches':>8s} {'wave_Mcyc':>9s} {'active':>7s} {'valu':>6s} {'lds':>6s} {'wait_any':>8s} {'wait_inst':>9s} {'wait_lds':>8s}")
for k, c in rows[:24]:
    w = c["SQ_WAVE_CYCLES"]
    if w <= 0:
        continue
    f = lambda x: f"{100 * c[x] / w:5.1f}%"
    print(f"{k[:44]:44s} {n[k]:8d} {4 * w / n[k] / 1e6:9.2f} {f('SQ_ACTIVE_INST_ANY'):>7s} {f('SQ_ACTIVE_INST_VALU'):>6s} "
          f"{f('SQ_ACTIVE_INST_LDS'):>6s} {f('SQ_WAIT_ANY'):>8s} {f('SQ_WAIT_INST_ANY'):>9s} {f('SQ_WAIT_INST_LDS'):>8s}")
