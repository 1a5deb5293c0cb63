"""Per-kernel MFMA utilisation from a rocprofv3 --pmc database written by scripts/pmc_mfma.sh.

    python scripts/pmc_mfma.py gpurun_out/<tag>_mfma/m_results.db > profiles/<tag>_mfma_busy.txt

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x elapsed shader cycles), the share of SIMD-cycles in
which a matrix instruction occupied the matrix pipe.  Elapsed cycles of a dispatch = GRBM_GUI_ACTIVE / 8
(rocprofv3 reports the sum over the 8 XCDs).  The counter counts cycles: 16 per v_mfma_f32_16x16x32_bf16,
32 per v_mfma_f32_32x32x16_bf16, so mfma_busy x 2.5 PFLOP/s x (clock / 2.4 GHz) is the bf16 rate it
corresponds to.  `ghz` is the effective clock of the profiled dispatch (elapsed cycles / duration)."""
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
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"])
print(f"{'kernel':44s} {'launches':>8s} {'us/launch':>9s} {'ghz':>5s} {'mfma_Mcyc':>9s} {'mfma_busy':>9s} {'sq_busy':>8s} {'bf16_mops':>10s}")
for k, c in rows[:24]:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    if cyc <= 0 or n[k] == 0:
        continue
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
    sqb = c["SQ_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"] if c["GRBM_GUI_ACTIVE"] else 0.0
    print(f"{k[:44]:44s} {n[k]:8d} {dur[k] / n[k] / 1e3:9.2f} {cyc / dur[k]:5.2f} {c['SQ_VALU_MFMA_BUSY_CYCLES'] / n[k] / 1e6:9.3f} "
          f"{100 * busy:8.1f}% {sqb:8.2f} {c['SQ_INSTS_VALU_MFMA_MOPS_BF16'] / n[k] / 1e6:10.3f}")
