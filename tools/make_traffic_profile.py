"""profiles/<tag>_traffic_b<B>.json from two rocprofv3 PMC passes of bench.py:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -o f -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --inflight 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <write_dir> -o w -- python bench.py ... (same)
usage: python tools/make_traffic_profile.py <fetch_dir> <write_dir> <batch> <out.json>

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE counts half the bytes of wide (16 B per lane) coalesced streams, so kernels
whose reads are such streams (stem, lin1, lin2, gate_last's table rows) get hbm = 2*FETCH + WRITE.  The
block-fused gate kernels (round 2) read their tables as 16-byte-per-lane LDS-DMA streams and their rows as 8 / 4
/ 2-byte words per lane; calibration on their own access pattern: block f4 at B = 256 must read at least its
input rows (256 x 28,672 B) and its tables once (2.5 MiB) = 9.96 MB, while raw FETCH_SIZE reports 6.45 MB -- so the
raw counter under-counts here too, and 2 x FETCH = 12.9 MB is the consistent reading (the surplus over 9.96 MB:
the Block_conv3 rows shared by a strand pair that miss L2, tables pulled into two XCDs).  They get the same
correction; both the raw and the corrected figure are kept.  The two-launch gate kernels of round 1 read narrow
words (uncalibrated) and stay uncorrected."""
import csv, glob, json, os, sys, collections

LABELS = [  # (substring of the kernel name, bench.py label, wide-stream correction)
    ("stem_pc_kernel", "stem", True), ("gate_last", "gate_last", True), ("gemm_f16x2_kernel", "head.lin1", True),
    ("lin2_f16x2_kernel", "head.lin2", True), ("head_mid_kernel", "head.bn_poly", False),
    # block-fused gate path (round 2): rows in as 8 / 4 / 2-byte words per lane, tables as 16-byte LDS-DMA streams
    ("gate_block_kernel<56, 29", "gate_block.f4", True), ("gate_block_kernel<29, 15", "gate_block.f5", True),
    ("gate_block_kernel<15, 8", "gate_block.f6", True), ("gate_block_kernel<8, 5", "gate_block.f7", True),
    # two-launch gate path (TTNET_GATE_UNFUSED=1, and the stride-1 blocks of --layers 3 / 4)
    ("gate_stage1_kernel<4, 4, 2, 2, 56, 29>", "gate_stage1.f4", False), ("gate_stage1_kernel<4, 4, 2, 2, 29, 15>", "gate_stage1.f5", False),
    ("gate_stage1_kernel<4, 4, 2, 2, 15, 8>", "gate_stage1.f6", False), ("gate_pf_kernel<29, 8>", "gate_pf.f4", False),
    ("gate_pf_kernel<15, 8>", "gate_pf.f5", False),
]

def per_launch(d, counter, batch_grid_hint=None):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}

fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
batch = int(sys.argv[3])
out = {"_doc": __doc__.strip().split("\n\n")[1].replace("\n", " "), "batch": batch, "kernels": {}}
GATE_ALGORITHMIC = 74592 * batch + 14155776
for sub, label, wide in LABELS:
    f = [v for k, v in fetch.items() if sub in k]
    w = [v for k, v in write.items() if sub in k]
    if not f or not w:
        continue
    out["kernels"][label] = {"fetch_kib": round(f[0], 1), "write_kib": round(w[0], 1),
                             "hbm_bytes": int(((2 if wide else 1) * f[0] + w[0]) * 1024), "fetch_x2_correction": wide,
                             "hbm_bytes_uncorrected": int((f[0] + w[0]) * 1024)}
gate = [v["hbm_bytes"] for k, v in out["kernels"].items() if k.startswith(("gate_block", "gate_stage1", "gate_pf"))]
if gate:
    out["gate_path"] = {"hbm_bytes": sum(gate), "algorithmic_bytes": GATE_ALGORITHMIC, "ratio": round(sum(gate) / GATE_ALGORITHMIC, 3)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
