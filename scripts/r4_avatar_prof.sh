# round 4, GPU box: evidence for the avatar-loss line (VERDICT r3 missing #4): bench line, kernel trace summary, SQ + HBM counters of the loss kernels.
# usage: bash scripts/r4_avatar_prof.sh TAG
tag=${1:-avatar}; out=$GRAFT_REPO_ROOT/gpurun_out/r4_$tag; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --loss avatar --steps 100 --warmup 5 --no-cpu-baseline 2> $out/bench.err | tail -1 > $out/bench_avatar_C3.json
python3 - $out/bench_avatar_C3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("avatar C3: %.1f frames/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
print({k: round(v, 1) for k, v in d["roofline"]["stage_us_per_step"].items()})
PY
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --loss avatar --steps 20 --warmup 5 --no-cpu-baseline --no-stage-timers > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_avatar_C3.csv
head -28 $out/kernel_stats_avatar_C3.csv | cut -c1-150
re='ssim|masked_l1|cos_loss|view_finish|frame_loss|mean_finish|avatar_'
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "$re" --output-format csv -d $out/pmc1 -- python3 bench.py --loss avatar --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > $out/pmc1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-include-regex "$re" --output-format csv -d $out/pmc2 -- python3 bench.py --loss avatar --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > $out/pmc2.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$re" --output-format csv -d $out/pmc3 -- python3 bench.py --loss avatar --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > $out/pmc3.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$re" --output-format csv -d $out/pmc4 -- python3 bench.py --loss avatar --steps 3 --warmup 2 --no-cpu-baseline --no-stage-timers > $out/pmc4.log 2>&1
python3 - $out > $out/counters_loss_kernels.txt <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("soar::(anonymous namespace)::", "").replace("void ", "")[:48]
            acc[(name, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"]
print("per launch (mean over the sampled launches; SQ counters in millions, FETCH_SIZE / WRITE_SIZE in MB as rocprofv3 reports them: KB / 1024)")
print("%-50s %-10s %5s " % ("kernel", "grid", "calls") + " ".join("%9s" % n.replace("SQ_", "")[:9] for n in names))
for (k, g), c in sorted(acc.items()):
    n = max(len(v) for v in c.values())
    vals = []
    for nm in names:
        v = c.get(nm)
        if not v: vals.append("%9s" % "-"); continue
        m = sum(v) / len(v)
        vals.append("%9.3f" % (m / 1024 if nm.endswith("_SIZE") else m / 1e6))
    print("%-50s %-10s %5d " % (k, g, n) + " ".join(vals))
PY
cat $out/counters_loss_kernels.txt | cut -c1-260
