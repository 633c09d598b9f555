#!/usr/bin/env python
"""Reduce the rocprofv3 --pmc passes of tools/pmc_round2.sh to per-kernel numbers (developer tool).

    python tools/pmc_summary2.py gpurun_out/<tag> profiles/<name>.json

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in separate passes, are in KiB, and on
gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream: bytes = 2 * FETCH_SIZE * 1024, WRITE_SIZE * 1024.
FETCH_SIZE counts fabric requests, Infinity-Cache hits included.  The third pass gives the average fabric read latency per kernel,
TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ (TCC clocks): reads served by the Infinity Cache return in roughly half the time of HBM misses,
so a kernel whose FETCH_SIZE exceeds its compulsory bytes but whose latency sits near the hit latency is re-reading on-die.
"""
import collections
import csv
import glob
import json
import re
import sys

PAT = re.compile(r"(gemm_pw2_kernel<\d, \d>|gemm_pw_kernel|rn_tail_kernel|gemm_kernel|rn_block128_kernel<\w+>|rn_sinc_kernel|res2net_chain_kernel|"
                 r"se_apply_kernel|asp_fused_kernel|fbank_kernel|rn_afms_apply_kernel|rn_maxpool3_kernel|se_mlp_kernel|rowvec_linear_kernel|"
                 r"colsum\w*_kernel|colstats_kernel|rn_afms_gate_kernel|rn_block_kernel|prologue\w*_kernel|pair_kernel<\d>|topk_stats\w*|l2norm_kernel)")


def key(row):
    n = row["Kernel_Name"]
    m = PAT.search(n)
    if not m and "rn_block128_kernel" in n:
        return "rn_block128_kernel"
    if not m:
        return None
    k = m.group(1)
    if k.startswith("gemm_pw2_kernel<2, 0>"):
        k += " grid=" + row["Grid_Size"]          # the K = 1024 layers and mfa are the same instance
    return k


def load(folder):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(folder + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r)
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    root, out = sys.argv[1:3]
    res = {"source": root, "corrections": "gfx950: bytes = 2 * FETCH_SIZE KiB * 1024 (wide coalesced reads); WRITE_SIZE KiB * 1024"}
    for model in ("ecapa", "rawnet2"):
        f, w, l = load(f"{root}/{model}_fetch"), load(f"{root}/{model}_write"), load(f"{root}/{model}_lat")
        tab = {}
        for k in sorted(set(f) | set(w) | set(l)):
            nf, nw = len(f[k]["FETCH_SIZE"]), len(w[k]["WRITE_SIZE"])
            rd, lev = sum(l[k]["TCC_EA0_RDREQ_sum"]), sum(l[k]["TCC_EA0_RDREQ_LEVEL_sum"])
            tab[k] = {"launches_in_pass": nf,
                      "fetch_bytes_per_launch": 2048.0 * sum(f[k]["FETCH_SIZE"]) / max(1, nf),
                      "write_bytes_per_launch": 1024.0 * sum(w[k]["WRITE_SIZE"]) / max(1, nw),
                      "fabric_read_latency_tcc_clk": lev / rd if rd else None}
            tab[k]["hbm_bytes_per_launch"] = tab[k]["fetch_bytes_per_launch"] + tab[k]["write_bytes_per_launch"]
        res[model] = tab
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
