# round 6, GPU box: which kernel of the KNN refresh takes 426 us per step at C5 (300k queries; 54 us at 100k)?
out=gpurun_out/r6_knn_c5; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o c5 -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stage-timers --workload C5 > $out/bench.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.reader(open(sys.argv[1])))[1:]
for r in rows[:18]: print(r[0][:90], r[1], "%.1f us avg" % (float(r[3])/1e3), "min %.1f max %.1f" % (float(r[5])/1e3, float(r[6])/1e3))
PY
