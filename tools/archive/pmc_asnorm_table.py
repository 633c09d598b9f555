"""Counters of the fused AS-norm kernels from the passes of tools/pmc_asnorm.sh:  python tools/pmc_asnorm_table.py gpurun_out/<tag>"""
import collections, csv, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(f"{root}/sq/p_counter_collection.csv")):
    k = r["Kernel_Name"]
    if "asnorm_fused" not in k and "cand_stats" not in k:
        continue
    k = "fused6" if "fused6" in k else "fused_f32" if "asnorm_fused_kernel" in k else "cand_stats"
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
for k, c in agg.items():
    d = dur[k]
    print(f"{k}: {n[k]} dispatches, {d / 1e6:.2f} ms under the profiler")
    print(f"  matrix pipe busy / (4 SIMDs x busy CU cycles) = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']):.3f}")
    print(f"  clock from SQ_BUSY_CU_CYCLES / 256 CUs / time = {c['SQ_BUSY_CU_CYCLES'] / 256 / d:.2f} GHz")
    print(f"  LDS array active {c['SQ_LDS_IDX_ACTIVE'] / c['SQ_BUSY_CU_CYCLES']:.3f} of the CU cycles, bank conflicts {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_BUSY_CU_CYCLES']:.3f}")
    print(f"  wave cycles: waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}, issue stalls {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}, issuing {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
