"""Per-kernel mean counter values from rocprofv3 --pmc runs stored as rocpd SQLite files.

    python tools/pmc_db.py <results.db> [<results.db> ...] [--match substring] [--json out.json]

Counters of one dispatch are summed over their instances (XCDs / shader engines); the mean is
over the dispatches of a kernel.  FETCH_SIZE / WRITE_SIZE are reported in bytes (KiB * 1024)."""
import collections
import json
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    return name.split("(")[0].split("<")[0][:48]


def main():
    args = sys.argv[1:]
    match, out = None, None
    if "--match" in args:
        i = args.index("--match"); match = args[i + 1]; del args[i:i + 2]
    if "--json" in args:
        i = args.index("--json"); out = args[i + 1]; del args[i:i + 2]
    table = collections.defaultdict(dict)
    for path in args:
        cur = sqlite3.connect(path).cursor()
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        durs = {}
        for name, disp, cname, val, dur in cur.execute(
                "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
            k = short(name)
            if match and match not in k:
                continue
            per[(k, cname)][disp] += val
            durs[(k, disp)] = dur
        for (k, cname), d in per.items():
            scale = 1024.0 if cname in ("FETCH_SIZE", "WRITE_SIZE") else 1.0
            table[k][cname] = scale * sum(d.values()) / len(d)
            table[k]["launches"] = len(d)
            table[k]["mean_ns_profiled"] = sum(durs[(k, x)] for x in d) / len(d)
    for k, row in sorted(table.items(), key=lambda kv: -kv[1].get("mean_ns_profiled", 0) * kv[1].get("launches", 0))[:40]:
        print(k, {c: (round(v) if v > 100 else round(v, 3)) for c, v in row.items()})
    if out:
        with open(out, "w") as f:
            json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
