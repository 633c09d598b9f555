"""per-kernel table of the passes of tools/pmc_scoring_r5.sh:  python tools/pmc_scoring_table.py gpurun_out/<tag>
FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes: MI355X_MICROARCH.md, HBM); both counters are in KiB."""
import collections, csv, glob, sys
root = sys.argv[1]
def short(k):
    for n in ("asnorm_h3w", "score_h3w", "asnorm_cand_stats", "pair_kernel", "split2_planes", "absmax_bits", "cohort_moments_part", "topk_stats"):
        if n in k:
            return n
    return None
def load(sub):
    f = glob.glob(f"{root}/{sub}/**/p_counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter(); seen = set()
for sub in ("sq", "fetch", "write"):
    for r in load(sub):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if sub == "sq" and (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k)); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
for k, c in agg.items():
    if not n[k]:
        continue
    d = dur[k]
    print(f"{k}: {n[k]} dispatches, {d / n[k] / 1e3:.1f} us each under the profiler")
    if c["SQ_BUSY_CU_CYCLES"]:
        print(f"  matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']):.3f} of the SIMD cycles; clock {c['SQ_BUSY_CU_CYCLES'] / 256 / d:.2f} GHz; "
              f"LDS active {c['SQ_LDS_IDX_ACTIVE'] / c['SQ_BUSY_CU_CYCLES']:.3f}, conflicts {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_BUSY_CU_CYCLES']:.3f}")
        print(f"  wave cycles: waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}, issue stalls {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}, issuing {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
    print(f"  fabric traffic per dispatch: fetched {2 * c['FETCH_SIZE'] * 1024 / n[k] / 1e6:.1f} MB (x2 corrected), written {c['WRITE_SIZE'] * 1024 / n[k] / 1e6:.1f} MB")
