# (the TCC block has 4 counter slots per pass: more than that aborts the profiler, which then hangs -- keep the timeout)
# L2 hit / miss counters of the kernels matching $1 (regex); further arguments KEY=VALUE are exported (development switches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
re="$1"; shift
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/pmcl2
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "$re" --output-format csv -d gpurun_out/pmcl2 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > gpurun_out/pmcl2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcl2/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r.get("Grid_Size", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k[0], "grid", k[1], k[2], "launches", len(v), "mean %.4g M" % (sum(v) / len(v) / 1e6))
PY
tail -3 gpurun_out/pmcl2.log | cut -c1-300
