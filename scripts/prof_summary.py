"""Per-kernel summary of a rocprofv3 --kernel-trace result database (rocpd sqlite).

    python scripts/prof_summary.py gpurun_out/prof_x/x_results.db [steps] > profiles/rNN_kernel_stats.csv

Prints name, calls, calls/step, total us, average us, share of kernel time, sorted by total."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").replace("_GLOBAL__N_1", "")
    m = re.match(r"_ZN2mi\d+([a-z0-9_]+_kernel)I(.*?)EEv", name)
    if m:
        args = m.group(2).replace("DF16b", "bf16,").replace("Li", "").replace("E", ",").replace("Lb", "")
        args = args.replace("f", "f32,", 1) if args.startswith("f") else args
        return f"{m.group(1)}<{args.strip(',')}>"
    return re.sub(r"\(.*", "", name).replace("mi::", "")


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("kernel,calls,calls_per_step,total_us,avg_us,percent")
    for name, n, t, a in rows:
        cps = f"{n / steps:.1f}" if steps else ""
        print(f"\"{short(name)}\",{n},{cps},{t / 1e3:.1f},{a / 1e3:.2f},{100.0 * t / tot:.2f}")
    if steps:
        print(f"\"TOTAL kernel time per step\",,,{tot / 1e3 / steps:.1f},,")


if __name__ == "__main__":
    main()
