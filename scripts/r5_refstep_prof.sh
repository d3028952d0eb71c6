out=gpurun_out/r5_refstep; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
SOAR_PROFILE_HOST=1 python scripts/refstep_time.py 2>&1 | grep -v "^$" | head -60
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/$out/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python3 $GRAFT_REPO_ROOT/scripts/refstep_time.py > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = 50   # 20 warm-up + 30 timed
print("GPU busy per step (all kernels / 50 steps): %.2f ms; launches per step %.0f" % (tot / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print("%-64s calls/step %5.1f  per step %7.1f us  avg %7.1f us" % (r["Name"][:64], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e3, float(r["AverageNs"]) / 1e3))
PY
