# round 4, GPU box: parity subset + default bench line.  usage: bash scripts/r4_quick.sh TAG [pytest -k expression]
tag=${1:-quick}; out=gpurun_out/r4_$tag; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py tests/test_plugin_gpu.py tests/test_training_gpu.py -x -q -m gpu ${2:+-k "$2"} > $out/tests.txt 2>&1
tail -4 $out/tests.txt
python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2> $out/bench.err | tail -1 > $out/bench_C3.json
python - $out/bench_C3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("C3: %.1f frames/s  %.3f ms/step  frac %.3f (%s %.1f us)" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"]))
print({k: round(v, 1) for k, v in d["roofline"]["stage_us_per_step"].items()})
PY
