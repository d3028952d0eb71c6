# GPU box: the KNN follower alone -- kernel trace + SQ counters.  usage: bash scripts/r4_knn_prof.sh TAG
tag=${1:-knnprof}; out=gpurun_out/r4_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/knn_follow_only.py
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o t -- python3 $R/scripts/knn_follow_only.py > /dev/null 2>&1
python3 - $R/$out/trace <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "knn" in r["Name"]: print("%8.1f us x %4s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:90]))
PY
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $R/$out/pmc -o p -- python3 $R/scripts/knn_follow_only.py 100000 1e-5 6 > /dev/null 2>&1
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
