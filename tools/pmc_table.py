"""Per-kernel pipe-counter table from the passes of tools/pmc_kernel.sh:  python tools/pmc_table.py gpurun_out/<tag> [name filter ...]"""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    calls = collections.Counter()
    seen = set()
    try:
        rows = csv.DictReader(open(path))
    except FileNotFoundError:
        return agg, dur, calls
    for r in rows:
        k = r["Kernel_Name"].replace("void ", "").replace("svhip::(anonymous namespace)::", "").replace("_ZN5svhip12_GLOBAL__N_1", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            calls[k] += 1
    return agg, dur, calls


def main():
    root = sys.argv[1]
    filt = sys.argv[2:]
    out = {}
    passes = {n: load(f"{root}/{n}/p_counter_collection.csv") for n in ("sq", "insts", "grbm", "fetch", "write")}
    names = sorted(passes["sq"][1], key=lambda k: -passes["sq"][1][k])
    for k in names:
        if filt and not any(f in k for f in filt):
            continue
        sq, d, c = passes["sq"][0][k], passes["sq"][1][k], passes["sq"][2][k]
        if not c or not sq.get("SQ_BUSY_CU_CYCLES"):
            continue
        rec = {"launches": c, "avg_us_under_profiler": d / c / 1e3,
               "mfma_busy_of_cu_cycles": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * sq["SQ_BUSY_CU_CYCLES"]),
               "clock_GHz_from_busy_cu_cycles": sq["SQ_BUSY_CU_CYCLES"] / 256 / d,
               "lds_active_of_cu_cycles": sq["SQ_LDS_IDX_ACTIVE"] / sq["SQ_BUSY_CU_CYCLES"],
               "lds_bank_conflict_of_cu_cycles": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_BUSY_CU_CYCLES"],
               "wave_cycles_waiting": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
               "wave_cycles_issue_stalled": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
               "wave_cycles_issuing": sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"]}
        ins, _, ci = passes["insts"][0].get(k, {}), None, passes["insts"][2].get(k, 0)
        if ci:
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_WAVES"):
                rec[n.lower() + "_per_launch"] = ins.get(n, 0.0) / ci
        g, gd = passes["grbm"][0].get(k, {}), passes["grbm"][1].get(k, 0)
        if gd:
            rec["clock_GHz_from_grbm"] = g.get("GRBM_GUI_ACTIVE", 0.0) / 8 / gd
        f, fc = passes["fetch"][0].get(k, {}), passes["fetch"][2].get(k, 0)
        if fc:
            rec["fetch_MB_per_launch_x2_corrected"] = 2 * f.get("FETCH_SIZE", 0.0) * 1024 / fc / 1e6      # FETCH_SIZE is in KiB; gfx950 counts half
        w, wc = passes["write"][0].get(k, {}), passes["write"][2].get(k, 0)
        if wc:
            rec["write_MB_per_launch"] = w.get("WRITE_SIZE", 0.0) * 1024 / wc / 1e6
        out[k] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
