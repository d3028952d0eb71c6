# usage (on the GPU box): bash scripts/prof_cmd.sh <tag> <python script> [args...]  -> gpurun_out/<tag>/ kernel stats
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 "$@" > $out/run.log 2>&1
tail -2 $out/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$out/trace/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f}us tot={float(r['TotalDurationNs'])/1e6:7.2f}ms")
PY
