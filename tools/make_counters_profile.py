"""profiles/<tag>_counters_small_b256.json from rocprofv3 --pmc passes of bench.py (tools/collect_counters.sh).
usage: python tools/make_counters_profile.py <out.json> <pmc_dir> [<pmc_dir> ...]

Per kernel and launch (mean over the launches of the pass): the raw SQ counters, summed over the chip by
rocprofv3, plus derived figures: instructions per wave (SQ_INSTS_* / SQ_WAVES), matrix-pipe busy cycles and VALU
wave-instructions per SIMD (the chip has 1,024 SIMDs; SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles, the other
SQ_*_CYCLES quad-cycles: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Utilisation = per-SIMD busy cycles / (the
kernel's duration in profiles/<tag>_kernel_stats_small.csv x the clock the chip holds, 1.5 - 2.1 GHz)."""
import csv, glob, json, os, sys, collections

LABELS = [("stem_pc_kernel", "stem"), ("gate_last", "gate_last"), ("gemm_f16x2_kernel", "head.lin1"),
          ("lin1_", "head.lin1"), ("lin2_f16x2_kernel", "head.lin2"), ("head_mid_kernel", "head.bn_poly"),
          ("head_tail", "head.tail"),
          ("gate_block_kernel<56, 29", "gate_block.f4"), ("gate_block_kernel<29, 15", "gate_block.f5"),
          ("gate_block_kernel<15, 8", "gate_block.f6"),
          ("full_pw_fast_kernel<2>", "full.pw_fast<2>"), ("full_pw_fast_kernel<1>", "full.pw_fast<1>"),
          ("full_dw_fast_kernel<5, 6>", "full.dw_fast<5,6>"), ("full_dw_fast_kernel<6, 5>", "full.dw_fast<6,5>"),
          ("full_pw_mfma_kernel<2, true>", "full.pw_fix<2>"), ("full_pw_mfma_kernel<1, true>", "full.pw_fix<1>"),
          ("full_dw_fix_kernel", "full.dw_fix")]

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
out = {"_doc": __doc__.strip().split("\n\n")[1].replace("\n", " "), "batch": int(os.environ.get("COUNTERS_BATCH", "256")), "kernels": {}}
for name, cs in acc.items():
    label = next((l for s, l in LABELS if s in name), None)
    if label is None:
        continue
    c = {k: v[0] / v[1] for k, v in cs.items()}
    e = dict(sorted((k, round(v, 1)) for k, v in c.items()))
    if c.get("SQ_WAVES"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_SALU"):
            if k in c:
                e[k.replace("SQ_INSTS_", "").lower() + "_per_wave"] = round(c[k] / c["SQ_WAVES"], 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # summed over the chip's 1,024 SIMDs, in shader cycles: divide by (kernel time x clock) for the matrix-pipe utilisation
        e["mfma_busy_cycles_per_simd"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0, 1)
    if "SQ_INSTS_VALU" in c:
        e["valu_wave_instructions_per_simd"] = round(c["SQ_INSTS_VALU"] / 1024.0, 1)
    e["kernel_name"] = name[:120]
    out["kernels"][label] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
