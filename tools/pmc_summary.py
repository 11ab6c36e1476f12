"""Per-kernel HBM traffic from two rocprofv3 counter passes over the same command.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <write_dir> -- python3 bench.py ...
    python tools/pmc_summary.py <fetch_dir> <write_dir> profiles/<name>.json

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  On gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads (16 bytes per lane, global loads and LDS-DMA buffer loads alike)
at 64 bytes each, so the fetched bytes of such kernels are twice the reported figure; WRITE_SIZE is
exact for 16-byte-per-lane stores.  Both the raw and the corrected figure are written; the
correction is only valid for kernels that read with 16-byte lanes (all the MFMA conv kernels do).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def variant(name):
    """Template instances that are different kernels in all but name keep a tag."""
    if "wgrad_wide_kernel" in name:
        return "<4x4s2>" if "WideCfg<true" in name else "<3x3>"
    m = re.search(r"enc1_bwd2_kernel<\d+, (\d)>", name)
    if m:
        return "<gram>" if m.group(1) == "2" else "<routed>"
    return ""


def short_name(name):
    tag = variant(name)
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.replace("_ZN12_GLOBAL__N_1", "").split("(")[0]
    return (name.split("<")[0] if not name[:1].isdigit() else name) + tag


def per_kernel_db(path, counter):
    """rocpd SQLite output (rocprofv3 of ROCm 7): counter values of a dispatch are summed over their
    instances (XCDs)."""
    import sqlite3
    cur = sqlite3.connect(path).cursor()
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for name, disp, val in cur.execute(
            "select kernel_name, dispatch_id, value from counters_collection where counter_name = ?", (counter,)):
        per[short_name(name)][disp] += val * 1024.0
    return {k: list(d.values()) for k, d in per.items()}


def per_kernel(directory, counter):
    dbs = glob.glob(os.path.join(directory, "*.db")) + glob.glob(os.path.join(directory, "*", "*.db"))
    if dbs:
        return per_kernel_db(max(dbs, key=os.path.getmtime), counter)
    path = max(glob.glob(os.path.join(directory, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.replace("_ZN12_GLOBAL__N_1", "").split("(")[0]
            acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    rows = {}
    for name in sorted(set(fetch) | set(write), key=lambda n: -(sum(fetch.get(n, [0])) + sum(write.get(n, [0])))):
        fv, wv = fetch.get(name, []), write.get(name, [])
        f_avg = sum(fv) / len(fv) if fv else 0.0
        w_avg = sum(wv) / len(wv) if wv else 0.0
        rows[name] = {"launches": max(len(fv), len(wv)), "fetch_raw_bytes_per_launch": round(f_avg),
                      "write_bytes_per_launch": round(w_avg),
                      "traffic_bytes_per_launch": round(2.0 * f_avg + w_avg)}
    with open(out, "w") as f:
        json.dump({"unit": "bytes per launch (mean over all launches of the kernel in the run)",
                   "correction": "traffic = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 wide-read tally)",
                   "kernels": rows}, f, indent=1)
    for name, r in list(rows.items())[:25]:
        print(f"{name[:56]:56s} n={r['launches']:4d} fetch_raw={r['fetch_raw_bytes_per_launch'] / 1e6:9.2f} MB "
              f"write={r['write_bytes_per_launch'] / 1e6:9.2f} MB traffic={r['traffic_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
