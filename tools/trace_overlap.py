"""Reads a rocprofv3 --kernel-trace CSV of `bench.py` (two batches in flight) and prints, over the middle half of the two
lanes' kernels: time per step, the share of time with 0 / 1 / 2+ kernels of the forward dispatched, the gap between consecutive
kernels of each lane (by kernel pair), each kernel's dispatch-to-end time (which includes waiting for CUs the other lane's kernel
holds), and a sample of the timeline.
usage: python tools/trace_overlap.py <kernel_trace.csv>   (rocprofv3 --kernel-trace --output-format csv -- python bench.py --steps 400 --windows 1)"""
import csv, re, sys, collections

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ttnet" in r["Kernel_Name"]]
def short(n):
    m = re.search(r"(stem_pc_kernel|gate_block_kernel<\d+|gate_last8|gemm_f16x2|head_mid|lin2_f16x2)", n)
    return m.group(1) if m else None
byq = collections.Counter(r["Queue_Id"] for r in rows if short(r["Kernel_Name"]))
serial_q = byq.most_common(1)[0][0]                   # the one-at-a-time leg and warm-up run on the default stream
lanes = [q for q in byq if q != serial_q]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows
            if r["Queue_Id"] in lanes and short(r["Kernel_Name"]))
n = len(ev); ev = ev[n // 4: 3 * n // 4]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
cov = collections.Counter(); d = 0; last = t0
for t, x in pts:
    cov[min(d, 2)] += t - last; last = t; d += x
tot = sum(cov.values()); steps = sum(1 for e in ev if e[2] == "stem_pc_kernel")
print("lanes: queues", lanes, "| steps", steps, "| %.1f us per step (under the tracer)" % ((t1 - t0) / 1e3 / steps),
      "| share of time with 0 / 1 / 2+ kernels dispatched: %.1f / %.1f / %.1f %%" % tuple(100.0 * cov[i] / tot for i in (0, 1, 2)))
for q in lanes:
    l = [e for e in ev if e[3] == q]
    gaps = collections.defaultdict(list)
    for a, b in zip(l, l[1:]): gaps[a[2] + " -> " + b[2]].append((b[0] - a[1]) / 1e3)
    print("queue", q)
    for k, v in gaps.items():
        v.sort(); print("  %-46s median gap %5.1f us  p90 %5.1f" % (k, v[len(v) // 2], v[int(len(v) * 0.9)]))
dur = collections.defaultdict(list)
for s, e, k, q in ev: dur[k].append((e - s) / 1e3)
print("dispatch -> end (us):")
for k, v in dur.items():
    v.sort(); print("  %-24s median %5.1f  p10 %5.1f  p90 %5.1f" % (k, v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]))
print("timeline sample (queue, kernel, start, end in us):")
base = ev[200][0]
for s, e, k, q in ev[200:232]: print("  ", q, "%-24s" % k, "%7.1f %7.1f" % ((s - base) / 1e3, (e - base) / 1e3))
