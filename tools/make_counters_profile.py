"""profiles/<tag>_counters_small_b256.json from rocprofv3 --pmc passes of bench.py (tools/collect_counters.sh).
usage: python tools/make_counters_profile.py <out.json> <pmc_dir> [<pmc_dir> ...]

Per kernel and launch (mean over the launches of the pass): the raw SQ counters, summed over the chip by
rocprofv3, plus derived figures: VALU instructions per wave (SQ_INSTS_VALU / SQ_WAVES), MFMA busy share
(SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES -- both summed over the chip's units, see MI355X_MICROARCH.md
"rocprofv3 PMC slots": SQ_*_CYCLES count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles)."""
import csv, glob, json, os, sys, collections

LABELS = [("stem_pc_kernel", "stem"), ("gate_last", "gate_last"), ("gemm_f16x2_kernel", "head.lin1"),
          ("lin1_", "head.lin1"), ("lin2_f16x2_kernel", "head.lin2"), ("head_mid_kernel", "head.bn_poly"),
          ("head_tail", "head.tail"),
          ("gate_block_kernel<56, 29", "gate_block.f4"), ("gate_block_kernel<29, 15", "gate_block.f5"),
          ("gate_block_kernel<15, 8", "gate_block.f6")]

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
out = {"_doc": __doc__.strip().split("\n\n")[1].replace("\n", " "), "batch": 256, "kernels": {}}
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
    if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        e["mfma_busy_over_sq_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], 4)
    e["kernel_name"] = name[:120]
    out["kernels"][label] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
