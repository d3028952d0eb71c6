# round 4, GPU box: per-kernel time of the default bench step (rocprofv3 kernel trace), microseconds per step.  usage: bash scripts/r4_trace_default.sh TAG
tag=${1:-trace}; out=$GRAFT_REPO_ROOT/gpurun_out/r4_$tag; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-stage-timers > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_C3.csv
tail -1 $out/trace.log | cut -c1-200
python3 - $out/kernel_stats_C3.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 45.0
tot = 0.0
for r in rows:
    us = float(r["TotalDurationNs"]) / 1e3 / steps
    tot += us
    if us >= 1.0:
        name = r["Name"].replace("soar::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        print("%-62s calls/step %5.1f  avg %7.1f us  per step %7.1f us" % (name, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, us))
print("sum of kernel time per step: %.1f us" % tot)
PY
python3 - $out/trace <<'PY'
import csv, glob, sys, statistics, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"] + ("" if "knn" not in r["Kernel_Name"] else " grid " + str(r.get("Grid_Size_X", r.get("Grid_Size", "?"))))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("medians (us) of the launches of a kernel:")
for k, v in sorted(d.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1])):
    if len(v) >= 20: print("  %-60s n %4d  median %7.1f  min %7.1f" % (k.replace("soar::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:42] + k[k.rfind(" grid"):] if " grid" in k else k.replace("soar::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], len(v), statistics.median(v), min(v)))
PY
python3 - $out/trace <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("soar::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("knn_certify")]
if len(marks) > 12:
    a, b = marks[-6], marks[-5]
    print("one step, launch by launch (start offset us, duration us, gap to the previous launch's end us):")
    t0, prev_end, busy = rows[a][0], rows[a][0], 0.0
    for s0, e0, n in rows[a:b]:
        print("  %8.1f %7.1f %6.1f  %s" % ((s0 - t0) / 1e3, (e0 - s0) / 1e3, (s0 - prev_end) / 1e3, n))
        busy += (e0 - s0) / 1e3
        prev_end = e0
    print("  step %.1f us, kernels %.1f us, gaps %.1f us over %d launches" % ((rows[b][0] - t0) / 1e3, busy, (rows[b][0] - t0) / 1e3 - busy, b - a))
PY
rm -rf $out/trace
