"""Per-kernel time per train step from a rocprofv3 kernel trace, restricted to the timed loop (the
longest run of steps, marked by the loss's forward kernel, less than 25 ms apart).
    python tools/prof_steps.py x.db [steps] [out.csv]"""
import collections
import csv
import sqlite3
import sys


def window_rows(path, steps):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = sorted(cur.execute(f"select start, end, {name} from kernels").fetchall())
    marks = [r[0] for r in rows if "loss_fwd_kernel" in r[2] or "loss_main_kernel" in r[2]]     # one per train step
    runs, run = [], [marks[0]]
    for a, b in zip(marks, marks[1:]):
        if b - a < 25e6:
            run.append(b)
        else:
            runs.append(run)
            run = [b]
    runs.append(run)
    run = max(runs, key=len)
    steps = min(steps, len(run) - 1)
    return [r for r in rows if run[-1 - steps] <= r[0] < run[-1]], steps


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rows, steps = window_rows(path, steps)
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        agg[n][0] += 1
        agg[n][1] += e - s
    total = sum(v[1] for v in agg.values())
    span = rows[-1][1] - rows[0][0]
    print(f"{steps} steps: {span / 1e6 / steps:.3f} ms per step, kernels {total / 1e6 / steps:.3f} ms per step")
    order = sorted(agg.items(), key=lambda kv: -kv[1][1])
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "CallsPerStep", "MsPerStep", "AverageUs", "Percentage"])
            for n, (c, t) in order:
                w.writerow([n, round(c / steps, 2), round(t / 1e6 / steps, 4), round(t / 1e3 / c, 2), round(100.0 * t / total, 2)])
    for n, (c, t) in order[:45]:
        print(f"{n[:70]:70s} calls/step={c / steps:6.2f} ms/step={t / 1e6 / steps:7.3f} avg_us={t / 1e3 / c:8.1f}")


if __name__ == "__main__":
    main()
