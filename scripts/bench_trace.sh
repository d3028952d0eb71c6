# kernel-trace stats of the default bench (no stage timers) -> prints the top kernels; gpurun_out/<tag>/
tag=${1:-trace}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-timers > $out/run.log 2>&1
tail -1 $out/run.log | cut -c1-200
python3 scripts/trace_busy.py $out
python3 - <<PY
import csv, glob
f = glob.glob("$out/trace/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:${2:-12}]:
    print(f"{r['Name'][:110]:110s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f}us tot={float(r['TotalDurationNs'])/1e6:7.2f}ms")
PY
