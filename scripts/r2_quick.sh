# quick GPU check: rasterizer parity tests + default bench summary; extra env assignments as arguments (e.g. SOAR_FWD_PIXEL=1)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2q}
shift
for kv in "$@"; do export "$kv"; done
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $out/pytest.log
python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 - <<PY
import json
d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print(d["roofline"]["stage_us"])
PY
