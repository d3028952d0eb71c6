# round 4, GPU box: the KNN follower's tests + the default bench line (2 runs).  usage: bash scripts/r4_knn.sh TAG
tag=${1:-knn}; out=gpurun_out/r4_$tag; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python -m pytest tests/test_lbs_gpu.py tests/test_training_gpu.py -x -q -m gpu > $out/tests.txt 2>&1
tail -4 $out/tests.txt
for r in 1 2; do
python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2> $out/bench.err | tail -1 > $out/bench_$r.json
python - $out/bench_$r.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("C3: %.1f frames/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
print({k: round(v, 1) for k, v in d["roofline"]["stage_us_per_step"].items()})
PY
done
