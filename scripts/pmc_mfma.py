"""Per-kernel MFMA utilisation from a rocprofv3 --pmc database written by scripts/pmc_mfma.sh.

    python scripts/pmc_mfma.py gpurun_out/<tag>_mfma/m_results.db [gpurun_out/<tag>_prof/k_results.db] > profiles/<tag>_mfma_busy.txt

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over every SIMD of the chip, the cycles in which a matrix instruction
occupies the matrix pipe: 16 per v_mfma_f32_16x16x32_bf16, 32 per v_mfma_f32_32x32x16_bf16 (checked here: the
3x3 conv kernels' count equals their MFMA count x 16 exactly, and SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 their FLOPs).
  mfma_busy      = MFMA cycles / (1024 SIMDs x launch duration x 2.4 GHz): share of the SIMD-cycles the chip has at
                   its nominal clock, i.e. the fraction of the 2.5 PFLOP/s dense bf16 peak.  Durations are taken
                   from the plain kernel trace (second database) when given, else from this counter run (slower).
  busy/gui       = MFMA cycles / (1024 x GRBM_GUI_ACTIVE / 8): the same share against the cycles the hardware
                   itself counted around the dispatch; reads low on dispatches of tens of microseconds, whose
                   counting window is longer than the kernel (`ghz` = that window's cycles / kernel duration)."""
import sqlite3
import sys
from collections import defaultdict

from prof_summary import short

db = sqlite3.connect(sys.argv[1])
acc = defaultdict(lambda: defaultdict(float))
dur = defaultdict(float)
n = defaultdict(int)
seen = set()
for name, counter, value, disp, start, end in db.execute(
        "select kernel_name, counter_name, value, dispatch_id, start, end from counters_collection"):
    k = short(name)
    acc[k][counter] += value
    if disp not in seen:
        seen.add(disp)
        n[k] += 1
        dur[k] += end - start
trace = {}
if len(sys.argv) > 2:
    for name, avg in sqlite3.connect(sys.argv[2]).execute("select name, avg(end-start) from kernels group by name"):
        trace[short(name)] = avg
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"])
print(f"{'kernel':44s} {'launches':>8s} {'us/launch':>9s} {'us(pmc)':>8s} {'mfma_Mcyc':>9s} {'mfma_busy':>9s} {'busy/gui':>8s} {'ghz':>5s} {'bf16_GFLOP':>10s}")
for k, c in rows[:24]:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    if cyc <= 0 or n[k] == 0:
        continue
    us_pmc = dur[k] / n[k] / 1e3
    us = trace.get(k, dur[k] / n[k]) / 1e3
    mcyc = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n[k]
    print(f"{k[:44]:44s} {n[k]:8d} {us:9.2f} {us_pmc:8.2f} {mcyc / 1e6:9.3f} {100 * mcyc / (1024 * us * 1e-6 * 2.4e9):8.1f}% "
          f"{100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0):7.1f}% {cyc / dur[k]:5.2f} "
          f"{c['SQ_INSTS_VALU_MFMA_MOPS_BF16'] * 512 / n[k] / 1e9:10.3f}")
