# sweep of a development environment variable: usage bash scripts/r2_sweep.sh <tag> <VAR> v1 v2 ...  (first value also runs the parity tests)
tag=$1; var=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
first=1
for v in "$@"; do
  export $var=$v
  if [ $first = 1 ]; then first=0; timeout 900 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py -m gpu -x -q > $out/pytest_$v.log 2>&1; echo "$var=$v pytest rc $?: $(tail -1 $out/pytest_$v.log)"; fi
  python bench.py --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err
  python3 - <<PY
import json
d = json.loads(open("$out/bench_$v.json").read().strip().splitlines()[-1])
su = d["roofline"]["stage_us"]
print("$var=$v: value", d["value"], "ms/step", d["ms_per_step"], "fwd", su["render_forward"], "bwd", su["render_backward"])
PY
done
