#!/usr/bin/env python
"""Reduce the rocprofv3 --pmc passes of tools/pmc_round3.sh (developer tool).

    python tools/pmc_summary3.py gpurun_out/<tag> profiles/<summary>.json [profiles/pmc_traffic.json]

Corrections as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: FETCH_SIZE and WRITE_SIZE come from separate passes, are
in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream: bytes = 2 * FETCH_SIZE * 1024,
WRITE_SIZE * 1024.  FETCH_SIZE counts fabric requests, Infinity-Cache hits included.  The optional third argument rewrites the
per-label table bench.py reads `roofline.traffic` from.
"""
import collections
import csv
import glob
import json
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from demangle import demangle  # noqa: E402

BASE = re.compile(r"(\w+_kernel)(<[^(]*>)?")


def label(name):
    name = demangle(name)
    name = re.sub(r"_ZN5svhip12_GLOBAL__N_1\d+", "", name.replace("svhip::(anonymous namespace)::", "").replace("void ", ""))
    name = re.sub(r"_kernelI.*", "_kernel", name) if name.startswith(("gemm", "rn_", "se_", "prologue")) and "<" not in name else name
    m = BASE.search(name)
    if not m:
        return None, None
    base, args = m.group(1), (m.group(2) or "")
    a = [x.strip() for x in args.strip("<>").split(",")] if args else []
    lb = base[:-len("_kernel")]
    if base == "gemm_pw3_kernel" and len(a) >= 4:
        cv = len(a) >= 5 and a[4] in ("true", "1")
        x3 = a[2] in ("true", "1")
        lb = "gemm_pw3r2" if a[3] in ("true", "1") else ("gemm_pw3cv" if cv else "gemm_pw3x3") if x3 else "gemm_pw3cv16" if cv else "gemm_pw3"
    elif base == "res2net_chain_kernel" and len(a) >= 2 and a[1] in ("3", "2"):
        lb = "res2net_slices"
    elif base == "gemm_pw2_kernel" and len(a) >= 3 and a[2] in ("true", "1"):
        lb = "gemm_pw2_conv"
    elif base == "gemm_kernel":
        lb = "gemm_generic"
    return lb, base + args


def load(folder):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for f in glob.glob(folder + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            lb, inst = label(r["Kernel_Name"])
            if lb:
                acc[lb][inst + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def reduce(f, w, l):
    nf, nw = len(f["FETCH_SIZE"]), len(w["WRITE_SIZE"])
    rd, lev = sum(l["TCC_EA0_RDREQ_sum"]), sum(l["TCC_EA0_RDREQ_LEVEL_sum"])
    e = {"launches_in_pass": nf, "fetch_bytes_per_launch": 2048.0 * sum(f["FETCH_SIZE"]) / max(1, nf),
         "write_bytes_per_launch": 1024.0 * sum(w["WRITE_SIZE"]) / max(1, nw), "fabric_read_latency_tcc_clk": lev / rd if rd else None}
    e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
    return e


def merge(d):
    out = collections.defaultdict(list)
    for inst in d.values():
        for k, v in inst.items():
            out[k] += v
    return out


def main():
    root, out = sys.argv[1:3]
    res = {"source": root, "corrections": "gfx950: bytes = 2 * FETCH_SIZE KiB * 1024 (wide coalesced reads); WRITE_SIZE KiB * 1024"}
    traffic = {"source": f"{out} (tools/pmc_round3.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py; gfx950 FETCH x2 correction)"}
    for model in ("ecapa", "rawnet2", "f32x3"):
        f, w, l = load(f"{root}/{model}_fetch"), load(f"{root}/{model}_write"), load(f"{root}/{model}_lat")
        tab = {}
        for lb in sorted(set(f) | set(w)):
            e = reduce(merge(f[lb]), merge(w[lb]), merge(l[lb]))
            insts = sorted(set(f[lb]) | set(w[lb]))
            if len(insts) > 1:
                e["per_instance"] = {i: reduce(f[lb][i], w[lb][i], l[lb][i]) for i in insts}
            tab[lb] = e
            traffic.setdefault(lb, e)
        res[model] = tab
    json.dump(res, open(out, "w"), indent=1)
    if len(sys.argv) > 3:
        json.dump(traffic, open(sys.argv[3], "w"), indent=1)
    for m in ("ecapa", "rawnet2", "f32x3"):
        for lb, e in res[m].items():
            print(f"{m:8s} {lb:22s} n={e['launches_in_pass']:4d} fetch {e['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {e['write_bytes_per_launch'] / 1e6:9.1f} MB  lat {e['fabric_read_latency_tcc_clk'] or 0:7.0f}")


if __name__ == "__main__":
    main()
