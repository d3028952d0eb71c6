# SQ counters of the kernels matching $1 (regex), default bench form; further arguments KEY=VALUE are exported (development switches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
re="$1"; shift
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/pmck1 gpurun_out/pmck2
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "$re" --output-format csv -d gpurun_out/pmck1 -- python3 bench.py $SOAR_PMC_BENCH_ARGS --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > gpurun_out/pmck1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-include-regex "$re" --output-format csv -d gpurun_out/pmck2 -- python3 bench.py $SOAR_PMC_BENCH_ARGS --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > gpurun_out/pmck2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmck1", "gpurun_out/pmck2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            # batched launches only (grid y = frames of a step)
            acc[(r["Kernel_Name"][:70], r.get("Grid_Size_Y", r.get("Grid_Size", "")), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(k[0], "gridY", k[1], k[2], "launches", len(v), "mean %.4g M" % (sum(v) / len(v) / 1e6))
PY
