"""Per-queue view of ONE steady-state step from a rocprofv3 kernel trace of the plan-mode bench: for every hardware queue the
kernels in launch order with the idle gap in front of each -- is a frame chain waiting for its own previous kernel, or for a slot?"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
knn = sorted(int(r["Start_Timestamp"]) for r in rows if "knn_cell_kernel" in r["Kernel_Name"])
lo, hi = knn[-3], knn[-2]
sel = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows
             if lo <= int(r["Start_Timestamp"]) < hi)
print("step span %.1f us, %d launches" % ((hi - lo) / 1e3, len(sel)))
byq = collections.defaultdict(list)
for s, e, n, q in sel:
    m = re.search(r"(\w+_kernel|\w+Kernel|fillBuffer\w*|copyBuffer\w*)", n)
    byq[q].append((s, e, (m.group(1) if m else n[:40])[:34]))
for q, ks in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in ks) / 1e3
    span = (ks[-1][1] - ks[0][0]) / 1e3
    print("queue %s: %d kernels, first start %.1f, span %.1f us, busy %.1f us, idle inside %.1f us" % (q, len(ks), (ks[0][0] - lo) / 1e3, span, busy, span - busy))
    prev = ks[0][0]
    line = []
    for s, e, n in ks:
        line.append("%s[gap %.0f run %.0f]" % (n.replace("_kernel", ""), (s - prev) / 1e3, (e - s) / 1e3))
        prev = e
    print("   " + " ".join(line))
