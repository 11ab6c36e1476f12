"""Two rocprofv3 kernel traces side by side: per-kernel total time of each, the difference, and the
GPU busy time (union of kernel intervals) of both.   python tools/prof_diff.py a.db b.db [steps]"""
import sqlite3
import sys

from prof_summary import from_db


def busy(path):
    cur = sqlite3.connect(path).cursor()
    iv = sorted(cur.execute("select start, end from kernels").fetchall())
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    tot += ce - cs
    return tot, iv[-1][1] - iv[0][0]


a, b = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ra = {r[0]: r for r in from_db(a)}
rb = {r[0]: r for r in from_db(b)}
ba, bb = busy(a), busy(b)
print(f"busy {ba[0] / 1e6 / steps:.3f} vs {bb[0] / 1e6 / steps:.3f} ms/step; sum of kernels {sum(r[2] for r in ra.values()) / 1e6 / steps:.3f} vs "
      f"{sum(r[2] for r in rb.values()) / 1e6 / steps:.3f}")
rows = []
for n in set(ra) | set(rb):
    ta = ra[n][2] if n in ra else 0
    tb = rb[n][2] if n in rb else 0
    rows.append((tb - ta, n, ta, tb, ra[n][1] if n in ra else 0, rb[n][1] if n in rb else 0))
for d, n, ta, tb, ca, cb in sorted(rows, key=lambda r: -abs(r[0]))[:25]:
    print(f"{n[:60]:60s} {ta / 1e6 / steps:7.3f} -> {tb / 1e6 / steps:7.3f} ms/step ({d / 1e6 / steps:+.3f})  calls {ca} -> {cb}")
