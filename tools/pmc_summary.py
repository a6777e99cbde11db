"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ttnet::", "").split("(")[0][:70]
        if sub in k:
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (v, n) in sorted(cs.items()):
        print(f"   {c:32s} {v / n:16.1f}  per launch ({n} launches)")
