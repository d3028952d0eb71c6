out=gpurun_out/r5_refstep; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python scripts/refstep_time.py 2>&1 | tail -1
SOAR_REFSTEP_LOSS=mean python scripts/refstep_time.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python3 $GRAFT_REPO_ROOT/scripts/refstep_time.py > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = 50   # 20 warm-up + 30 timed
print("GPU busy per step (all kernels / 50 steps): %.2f ms" % (tot / steps / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-60s calls %6s  total %8.2f ms  avg %7.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
