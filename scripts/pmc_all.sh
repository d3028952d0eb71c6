# instruction mix of every kernel of a serialized (plan-eager) step: VALU / SALU / LDS / VMEM / SMEM instructions and wave-cycles per launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmca -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers --mode plan-eager > gpurun_out/pmca.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmca/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        rows.append((m.get("SQ_INSTS_VALU", 0) + m.get("SQ_INSTS_SALU", 0), k, len(c["SQ_WAVES"]), m))
    print("%-64s %5s %8s %8s %8s %8s %8s %8s" % ("kernel", "calls", "waves", "VALU", "SALU", "LDS", "VMEM", "SMEM"))
    for _, k, n, m in sorted(rows, reverse=True)[:24]:
        print("%-64s %5d %8.0f %8.3g %8.3g %8.3g %8.3g %8.3g" % (k, n, m.get("SQ_WAVES", 0), m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0),
              m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_VMEM", 0), m.get("SQ_INSTS_SMEM", 0)))
PY
