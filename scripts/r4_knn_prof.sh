# GPU box: the KNN refresh alone (scripts/knn_follow_only.py) -- kernel trace medians per (kernel, grid), optionally SQ counters.
# usage: bash scripts/r4_knn_prof.sh TAG [pmc]      (SOAR_HIP_LIB selects a variant library)
tag=${1:-knnprof}; out=gpurun_out/r4_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/knn_follow_only.py | tail -1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace -o t -- python3 $R/scripts/knn_follow_only.py > /dev/null 2>&1
python3 - $R/$out/trace <<'PY'
import csv, glob, sys, statistics, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "knn" in r["Kernel_Name"]:
            d[(r["Kernel_Name"].replace("soar::(anonymous namespace)::", "").split("(")[0][:40], r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(d.items()):
    if len(v) >= 10: print("  %-42s grid %8s  n %3d  median %6.1f us  min %6.1f" % (k, g, len(v), statistics.median(v), min(v)))
PY
rm -rf $R/$out/trace
[ "$2" = "pmc" ] || exit 0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $R/$out/pmc -o p -- python3 $R/scripts/knn_follow_only.py 100000 2e-6 6 > /dev/null 2>&1
python3 - $R/$out/pmc <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "knn_certify" in k or "knn_blend" in k: acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v[-4:]) / len(v[-4:]) / 1e6, 3) for c, v in d.items()}, "(millions per launch)")
PY
rm -rf $R/$out/pmc
done
