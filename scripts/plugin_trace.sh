# gpurun -- 'bash scripts/plugin_trace.sh': kernel trace of the plugin-level training loop (scripts/plugin_time.py)
out=$GRAFT_REPO_ROOT/gpurun_out/plugin_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/plugin_time.py > $out/log.txt 2>&1
tail -2 $out/log.txt
python3 scripts/trace_busy.py $out
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms {100 * float(r["TotalDurationNs"]) / tot:5.1f}%  calls {r["Calls"]:>6}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:110]}')
PY
