"""Diagnostic: summarise a SOAR_BWD_TIMELINE_FILE dump of the entry-lane backward blend (variant built with -DSOAR_BWD_TIMELINE):
per wavefront {t_start, t_prologue_done, t_first_chunk_staged, t_end (100 MHz wall clock), list length, deepest | deepest_wave << 32,
batches, pixel iterations}."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] > 0]
t0 = int(a[:, 0].min())
us = lambda col: (a[:, col].astype(np.int64) - t0) / 100.0
start, pro, staged, end = us(0), us(1), us(2), us(3)
ok = a[:, 1] > 0
print("wavefronts logged", len(a), "with work", int(ok.sum()), "launch span us %.1f" % end.max())
start, pro, staged, end, b = start[ok], pro[ok], staged[ok], end[ok], a[ok]
q = lambda v: "mean %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (v.mean(), *np.percentile(v, [10, 50, 90, 99]), v.max())
print("lifetime us        ", q(end - start))
print("prologue us        ", q(pro - start))
print("first chunk staged ", q(staged - pro))
print("walk us            ", q(end - staged))
batches, iters = b[:, 6].astype(float), b[:, 7].astype(float)
deep = (b[:, 5] & np.uint64(0xFFFFFFFF)).astype(float)
print("list length        ", q(b[:, 4].astype(float)))
print("deepest            ", q(deep))
print("batches / wavefront", q(batches), " pixel iterations", q(iters))
print("sum of lifetimes %.0f us = %.1f wavefronts in flight on average of %d slots" % ((end - start).sum(), (end - start).sum() / end.max(), 256 * 16))
for cut in np.linspace(0, end.max(), 12)[1:-1]:
    print("  t = %6.1f us: %5d wavefronts alive, %5d of them in prologue / first staging" % (cut, int(((start < cut) & (end > cut)).sum()), int(((start < cut) & (staged > cut)).sum())))
# walk time against work
w = end - staged
print("walk us per pixel iteration (wavefronts with >= 32 iterations): %.3f" % (w[iters >= 32].sum() / iters[iters >= 32].sum()))
print("walk us, wavefronts without any batch: ", q(w[batches == 0]) if (batches == 0).any() else "-")
