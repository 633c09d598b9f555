#!/usr/bin/env python
"""Per-kernel effective clock and matrix-pipe share from the two passes of tools/pmc_clock_r6.sh (developer tool).

    python tools/pmc_clock_table.py gpurun_out/<tag>

effective clock = GRBM_GUI_ACTIVE / 8 / (kernel end - start)   (the counter is summed over the 8 XCDs; it reads high on dispatches
shorter than ~0.3 ms: MI355X_MICROARCH.md "DVFS give-back").  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES-equivalent):
reported as busy cycles per SIMD (1 024 SIMDs) over the kernel's cycles at the effective clock of pass 1.
"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary3 import label  # noqa: E402


def load(folder):
    """-> {dispatch_id: (label, instance, grid, {counter: value}, duration_ns)}"""
    dur = {}
    for f in glob.glob(folder + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = {}
    for f in glob.glob(folder + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            lb, inst = label(r["Kernel_Name"])
            if not lb:
                continue
            d = r["Dispatch_Id"]
            e = rows.setdefault(d, [lb, inst, r["Grid_Size"], {}, dur.get(d)])
            if e[4] is None and "Start_Timestamp" in r:
                e[4] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e[3][r["Counter_Name"]] = e[3].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return rows


def group(rows):
    g = collections.defaultdict(list)
    for lb, inst, grid, c, d in rows.values():
        if d:
            g[(lb, inst, grid)].append((c, d))
    return g


def main():
    root = sys.argv[1]
    g1, g2 = group(load(root + "/grbm")), group(load(root + "/sq"))
    print(f"# {root}: per kernel instance, means over the launches of the pass (bench.py --steps 20 --warmup 5, bf16 ECAPA B = 256)")
    print(f"{'kernel':<58} {'n':>4} {'us':>8} {'clk GHz':>8} | {'us(sq)':>8} {'mfma busy':>9} {'wave act':>8} {'wait':>6} {'stall':>6} {'issue':>6}")
    keys = sorted(g1, key=lambda k: -sum(d for _, d in g1[k]))
    for k in keys[:24]:
        a = g1[k]
        n = len(a)
        us = sum(d for _, d in a) / n / 1e3
        clk = sum(c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / d for c, d in a) / n          # cycles per ns = GHz
        line = f"{(k[0] + ' ' + k[1])[:58]:<58} {n:>4} {us:>8.1f} {clk:>8.3f} |"
        b = g2.get(k)
        if b:
            m = len(b)
            us2 = sum(d for _, d in b) / m / 1e3
            cyc = clk * us2 * 1e3                                                         # shader cycles of the kernel at pass 1's clock
            mf = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c, _ in b) / m
            wc = sum(c.get("SQ_WAVE_CYCLES", 0.0) for c, _ in b) / m
            wa = sum(c.get("SQ_WAIT_ANY", 0.0) for c, _ in b) / m
            wi = sum(c.get("SQ_WAIT_INST_ANY", 0.0) for c, _ in b) / m
            ai = sum(c.get("SQ_ACTIVE_INST_ANY", 0.0) for c, _ in b) / m
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over 1 024 SIMDs (guide: = 32 x N_mfma for 32x32x16, in cycles)
            line += f" {us2:>8.1f} {mf / (1024.0 * cyc) if cyc else 0:>9.3f} {wc * 4 / (1024.0 * cyc) if cyc else 0:>8.2f} {wa / wc if wc else 0:>6.2f} {wi / wc if wc else 0:>6.2f} {ai / wc if wc else 0:>6.2f}"
        print(line)
    print("# mfma busy = SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x kernel cycles); wave act = SQ_WAVE_CYCLES x 4 (quad-cycles) / the same: waves resident per SIMD;")
    print("# wait / stall / issue = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES")


if __name__ == "__main__":
    main()
