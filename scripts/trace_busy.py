"""From a rocprofv3 kernel trace: wall span, union of kernel intervals (GPU busy) and sum of durations (overlap)."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
iv = np.array(sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows), dtype=np.int64)
names = [r["Kernel_Name"] for r in rows]
# steady state = from the 5th to the last KNN launch (one per optimizer step)
knn = sorted(int(r["Start_Timestamp"]) for r in rows if "knn_cell_kernel" in r["Kernel_Name"])
lo, hi, nsteps = knn[4], knn[-1], len(knn) - 5
iv = iv[(iv[:, 0] >= lo) & (iv[:, 0] < hi)]
span = hi - lo
print("steps in window:", nsteps, " ms/step %.3f" % (span / 1e6 / nsteps))
busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("steady-state span %.2f ms  busy(union) %.2f ms (%.1f%%)  sum of kernel durations %.2f ms  launches %d" %
      (span / 1e6, busy / 1e6, 100.0 * busy / span, (iv[:, 1] - iv[:, 0]).sum() / 1e6, len(iv)))
gaps = iv[1:, 0] - np.maximum.accumulate(iv[:-1, 1])
gaps = gaps[gaps > 0]
print("idle gaps: n=%d total %.2f ms, p50 %.1f us p90 %.1f us max %.1f us" % (len(gaps), gaps.sum() / 1e6, np.percentile(gaps, 50) / 1e3, np.percentile(gaps, 90) / 1e3, gaps.max() / 1e3))
