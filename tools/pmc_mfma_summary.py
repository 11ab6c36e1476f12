"""Per-kernel MFMA utilisation, effective clock, wait fractions and LDS bank conflicts from rocprofv3
counter passes over one command (rocpd SQLite output), one pass per run:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              GRBM_GUI_ACTIVE -d <dirA> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d <dirB> -- python3 bench.py ...
    python tools/pmc_mfma_summary.py profiles/<name>.json <dirA> <dirB>

Counters of one dispatch are summed over their instances (rocprofv3 reports one row per XCD /
shader engine); a kernel's figures are means over its dispatches in the run.

  eff_clock_ghz   = GRBM_GUI_ACTIVE / 8 / dispatch duration        (the counter is the sum over the 8 XCDs;
                    reads high on dispatches shorter than ~0.3 ms)
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
                    (busy cycles are summed over all SIMDs: 32 per v_mfma_f32_32x32x16_bf16)
  wait_any_frac   = SQ_WAIT_ANY / SQ_WAVE_CYCLES        waves parked in s_waitcnt / barriers
  wait_inst_frac  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   issue stalls (MFMA dependencies, busy pipes)
  active_frac     = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import glob
import json
import os
import re
import sqlite3
import sys

SIMDS = 1024


def variant(name):
    """Template instances that are different kernels in all but name keep a tag."""
    if "wgrad_wide_kernel" in name:
        return "<4x4s2>" if "WideCfg<true" in name else "<3x3>"
    m = re.search(r"enc1_bwd2_kernel<\d+, (\d)>", name)
    if m:
        return "<gram>" if m.group(1) == "2" else "<routed>"
    return ""


def short(name):
    tag = variant(name)
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    return name.split("(")[0].split("<")[0] + tag


def read_db(path, table):
    cur = sqlite3.connect(path).cursor()
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    durs = {}
    for name, disp, cname, val, dur in cur.execute(
            "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
        k = short(name)
        per[(k, cname)][disp] += val
        durs[(k, disp)] = dur
    for (k, cname), d in per.items():
        row = table[k]
        row[cname] = sum(d.values()) / len(d)
        row.setdefault("launches", len(d))
        ns = sum(durs[(k, x)] for x in d) / len(d)
        if cname == "GRBM_GUI_ACTIVE" or "mean_ns" not in row:
            row["mean_ns"] = ns                      # the duration of the pass that holds the clock counter


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    table = collections.defaultdict(dict)
    for d in dirs:
        dbs = [d] if d.endswith(".db") else glob.glob(os.path.join(d, "*.db")) + glob.glob(os.path.join(d, "*", "*.db"))
        read_db(max(dbs, key=os.path.getmtime), table)
    rows = {}
    order = sorted(table.items(), key=lambda kv: -kv[1].get("mean_ns", 0) * kv[1].get("launches", 0))
    for k, c in order:
        r = {"launches": c.get("launches", 0), "mean_us_profiled": round(c.get("mean_ns", 0) / 1e3, 2)}
        gui, wave = c.get("GRBM_GUI_ACTIVE"), c.get("SQ_WAVE_CYCLES")
        if gui and c.get("mean_ns"):
            r["eff_clock_ghz"] = round(gui / 8.0 / c["mean_ns"], 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                r["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * gui / 8.0), 4)
        if wave:
            for key, cname in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_frac", "SQ_WAIT_INST_ANY"),
                               ("active_frac", "SQ_ACTIVE_INST_ANY")):
                if cname in c:
                    r[key] = round(c[cname] / wave, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        r["counters"] = {n: round(v) for n, v in c.items() if n not in ("launches", "mean_ns")}
        rows[k] = r
    with open(out, "w") as f:
        json.dump({"command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (one rocprofv3 --pmc pass per counter group)",
                   "note": "means over the dispatches of a kernel in the run; durations are those of the counter pass "
                           "(serialised dispatches, slower than an unprofiled run)",
                   "formulas": {"eff_clock_ghz": "GRBM_GUI_ACTIVE / 8 / duration",
                                "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)",
                                "wait_any_frac": "SQ_WAIT_ANY / SQ_WAVE_CYCLES",
                                "lds_conflict_frac": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"},
                   "kernels": rows}, f, indent=1)
    for k, r in list(rows.items())[:30]:
        print(f"{k[:44]:44s} n={r['launches']:4d} us={r['mean_us_profiled']:9.1f} clk={r.get('eff_clock_ghz', 0):5.2f} "
              f"mfma={r.get('mfma_busy_frac', 0):6.3f} wait={r.get('wait_any_frac', 0):6.3f} "
              f"stall={r.get('wait_inst_frac', 0):6.3f} ldsconf={r.get('lds_conflict_frac', 0):6.3f}")


if __name__ == "__main__":
    main()
