# which kernels send atomics to the memory side, and how many requests their L2s see: gpurun -- 'bash scripts/pmc_atomics.sh [bench args]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rm -rf gpurun_out/tcca
timeout 120 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_REQ_sum --output-format csv -d gpurun_out/tcca -- python3 bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > gpurun_out/tcca.log 2>&1 || echo failed
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/tcca/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_REQ_sum": n[r["Kernel_Name"][:70]] += 1
    rows = sorted(acc.items(), key=lambda kv: -kv[1]["TCC_EA0_ATOMIC_sum"])
    for k, v in rows[:25]:
        print("%-70s launches %4d  EA atomics %8.3f M  L2 requests %8.3f M" % (k, n[k], v["TCC_EA0_ATOMIC_sum"] / 1e6, v["TCC_REQ_sum"] / 1e6))
PY
