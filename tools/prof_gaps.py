"""Idle gaps of the GPU in a rocprofv3 kernel trace: the largest gaps between consecutive kernels (by
the kernels on either side), and busy / idle time per step.   python tools/prof_gaps.py x.db steps [min_us]"""
import collections
import sqlite3
import sys

path, steps = sys.argv[1], int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
cur = sqlite3.connect(path).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = sorted(cur.execute(f"select start, end, {name} from kernels").fetchall())
# one loss_main_kernel per train step: the longest run of steps less than 25 ms apart is the timed loop;
# the window runs from the loss kernel `steps` steps before its last one to that last one
marks = [r[0] for r in rows if "loss_main_kernel" in r[2]]
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
rows = [r for r in rows if run[-1 - steps] <= r[0] < run[-1]]
gaps = collections.defaultdict(lambda: [0, 0.0])
busy_end = rows[0][1]
idle = 0.0
for (s, e, n), (ps, pe, pn) in zip(rows[1:], rows[:-1]):
    if s > busy_end:
        g = (s - busy_end) / 1e3
        idle += g
        if g >= min_us:
            k = (pn[:48], n[:48])
            gaps[k][0] += 1
            gaps[k][1] += g
    busy_end = max(busy_end, e)
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"{steps} steps, {span / 1e3 / steps:.3f} ms per step, idle {idle / 1e3 / steps:.3f} ms per step ({100 * idle / span:.1f} %)")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t / 1e3 / steps:7.3f} ms/step in {c:4d} gaps  after {a}  before {b}")
