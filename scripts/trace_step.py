"""Print the kernel timeline of ONE steady-state optimizer step from a rocprofv3 kernel trace (between two KNN launches)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
knn = sorted(int(r["Start_Timestamp"]) for r in rows if "knn_cell_kernel" in r["Kernel_Name"])
lo, hi = knn[-3], knn[-2]
sel = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows
             if lo - 200000 <= int(r["Start_Timestamp"]) < hi - 200000)
t0 = sel[0][0]
print("step span %.1f us, %d launches" % ((hi - lo) / 1e3, len(sel)))
for s, e, n, q in sel:
    m = re.search(r"(\w+_kernel|\w+Kernel|fillBuffer\w*|copyBuffer\w*)", n)
    short = m.group(1) if m else n[:40]
    if "rocprim" in n:
        short = "rocprim:" + short
    print("%8.1f  +%7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short[:60]))
