#!/usr/bin/env python
"""Reduce rocprofv3 --pmc passes to per-kernel HBM traffic (developer tool).

    python tools/pmc_summary.py <dir with FETCH pass> <dir with WRITE pass> profiles/pmc_traffic.json

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
read stream, so  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  per launch.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

LABELS = {"gemm_pw2_kernel": "gemm_pw2", "gemm_pw_kernel": "gemm_pw", "res2net_chain_kernel": "res2net_chain",
          "asp_fused_kernel": "asp_fused", "fbank_kernel": "fbank", "gemm_kernel": "gemm_generic",
          "se_apply_kernel": "se_apply", "colstats_kernel": "colstats"}


def label(name):
    if "gemm_pw2_kernel" in name:                   # gemm_pw2_kernel<EPI, M16, CONV, PH4>: CONV = true is the conv-gather instance
        args = name.replace(" ", "").split("<", 1)[-1].split(">", 1)[0].split(",")
        if len(args) >= 3 and args[2] in ("true", "1"):
            return "gemm_pw2_conv"
    for k, v in LABELS.items():
        if k in name:
            return v
    return None


def read_counter(folder, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            lb = label(row.get("Kernel_Name", ""))
            if lb:
                acc[lb][0] += float(row["Counter_Value"])
                acc[lb][1] += 1
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = read_counter(fdir, "FETCH_SIZE"), read_counter(wdir, "WRITE_SIZE")
    res = {}
    for lb in sorted(set(fe) | set(wr)):
        f = fe[lb][0] / max(1, fe[lb][1])
        w = wr[lb][0] / max(1, wr[lb][1])
        res[lb] = {"FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w, "launches_fetch_pass": fe[lb][1],
                   "launches_write_pass": wr[lb][1], "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                   "correction": "gfx950: FETCH_SIZE x2 (wide coalesced reads), WRITE_SIZE x1, KiB -> bytes"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
