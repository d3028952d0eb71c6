# SQ counters of the forward blend only, serialized launches (plan-eager)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "render_forward" --output-format csv -d gpurun_out/pmcf1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers --mode plan-eager > gpurun_out/pmcf1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-include-regex "render_forward" --output-format csv -d gpurun_out/pmcf2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers --mode plan-eager > gpurun_out/pmcf2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmcf1", "gpurun_out/pmcf2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(k[0], k[1], "launches", len(v), "mean %.4g" % (sum(v) / len(v)))
PY
