"""Diagnostic: summarise a SOAR_WAVE_LOG dump (per-wave {t_start, t_end, list length, blend iterations}, 100 MHz wall clock)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 4)
t_ready = ((a[..., 2] >> np.uint64(24)) & np.uint64(0xFFFFF)).astype(np.float64) / 100.0     # us from the start to the first staged chunk
t_tail = ((a[..., 2] >> np.uint64(44)) & np.uint64(0xFFFFF)).astype(np.float64) / 100.0      # us from the end of the blending to the end
a[..., 2] &= np.uint64((1 << 24) - 1)
t0 = a[..., 0][a[..., 0] > 0].min()
start = (a[..., 0].astype(np.int64) - int(t0)) / 100.0      # us
end = (a[..., 1].astype(np.int64) - int(t0)) / 100.0
dur = end - start
print("waves", dur.size, "kernel span us", end.max(), "first start", start.min(), "last start", start.max())
print("wave duration us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
order = np.argsort(dur.ravel())[::-1][:15]
for o in order:
    t, w = divmod(o, 4)
    print(f"tile {t} wave {w}: start {start.ravel()[o]:8.1f} end {end.ravel()[o]:8.1f} dur {dur.ravel()[o]:8.1f} us  list {a[t, w, 2]}  iters {a[t, w, 3]}")
for cut in (25, 50, 100, 150, 200, 250, 300):
    print(f"waves still running at {cut} us:", int(((start < cut) & (end > cut)).sum()))
useful = (a[..., 3] >> np.uint64(24)).astype(np.float64)
a[..., 3] &= np.uint64((1 << 24) - 1)
it = a[..., 3].astype(np.float64)
ln = a[..., 2].astype(np.float64)
print("sum iters %.3e  (per SIMD: %.0f)   sum list entries over waves %.3e (phase-A sub-chunk tests per SIMD: %.0f)" %
      (it.sum(), it.sum() / 1024, ln.sum(), ln.sum() / 64 / 1024))
busy = (a[..., 2] > 0)
print("waves with a non-empty list:", int(busy.sum()), " iters/wave among them mean %.1f max %d" % (it[busy].mean(), it.max()))
print("useful (pixel, entry) pairs: %.3e of %.3e evaluated lanes = %.1f %%" % (useful.sum(), it.sum() * 64, 100 * useful.sum() / max(it.sum() * 64, 1)))
# where a wavefront's time goes: least squares of its duration against its list length (staging + phase A) and its blend steps
sel = busy.ravel()
A = np.stack([np.ones(sel.sum()), ln.ravel()[sel], it.ravel()[sel]], axis=1)
coef, *_ = np.linalg.lstsq(A, dur.ravel()[sel], rcond=None)
res = dur.ravel()[sel] - A @ coef
print("duration ~ %.2f us + %.2f ns per list entry + %.1f ns per blend step (4 entries x 16 pixels); residual rms %.2f us; mean list %.0f, mean steps %.1f"
      % (coef[0], 1e3 * coef[1], 1e3 * coef[2], float(np.sqrt((res ** 2).mean())), ln.ravel()[sel].mean(), it.ravel()[sel].mean()))
print("of a wavefront's life: start -> first chunk staged %.2f us (p50 %.2f, p90 %.2f); end of blending -> end %.2f us (p50 %.2f, p90 %.2f)"
      % (t_ready.ravel()[sel].mean(), np.percentile(t_ready.ravel()[sel], 50), np.percentile(t_ready.ravel()[sel], 90),
         t_tail.ravel()[sel].mean(), np.percentile(t_tail.ravel()[sel], 50), np.percentile(t_tail.ravel()[sel], 90)))
