"""Per-kernel summary of a rocprofv3 --kernel-trace run: reads the rocpd SQLite file (or a
kernel_stats CSV) and prints / writes calls, total and average duration per kernel name.

    python tools/prof_summary.py gpurun_out/prof_x/x_results.db [steps] [out.csv]
"""
import csv
import sqlite3
import sys


def from_db(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       f"from kernels group by {name} order by 3 desc").fetchall()
    return rows


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = from_db(path)
    total = sum(r[2] for r in rows)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for n, c, t, a, mn, mx in rows:
                w.writerow([n, c, t, round(a, 1), round(100.0 * t / total, 2), mn, mx])
    print(f"total kernel time {total / 1e6:.2f} ms = {total / 1e6 / steps:.3f} ms per step over {steps:g} steps")
    for n, c, t, a, mn, mx in rows[:60]:
        print(f"{n[:64]:64s} calls={c:5d} per_step_ms={t / 1e6 / steps:7.3f} avg_us={a / 1e3:8.1f}")


if __name__ == "__main__":
    main()
