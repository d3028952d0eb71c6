# L2 (TCC) counters of the kernels matching $1 (regex), default bench form, two counters per pass (more "exceeds the capabilities of
# the hardware to collect"): gpurun -- 'bash scripts/pmc_tcc.sh render_backward'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
re="$1"; shift
for kv in "$@"; do export "$kv"; done
i=0
for pair in "TCC_HIT_sum TCC_MISS_sum" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_PROBE_sum TCC_REQ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1)); rm -rf gpurun_out/tcc$i
  timeout 90 rocprofv3 --pmc $pair --kernel-include-regex "$re" --output-format csv -d gpurun_out/tcc$i -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > gpurun_out/tcc$i.log 2>&1 || echo "pass $i ($pair) failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/tcc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r.get("Grid_Size_Y", r.get("Grid_Size", "")), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k[0], "gridY", k[1], k[2], "launches", len(v), "mean %.4g M" % (sum(v) / len(v) / 1e6))
PY
