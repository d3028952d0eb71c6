# round 4, GPU box: the avatar-loss path's tests + the avatar bench line (3 runs).  usage: bash scripts/r4_avatar.sh TAG
tag=${1:-avatar}; out=gpurun_out/r4_$tag; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python -m pytest tests/test_plugin_gpu.py tests/test_training_gpu.py tests/test_bench_gpu.py -x -q -m gpu -k "avatar or loss or fused_view or training or plan" > $out/tests.txt 2>&1
tail -6 $out/tests.txt
for r in 1 2 3; do
python bench.py --loss avatar --steps 100 --warmup 5 --no-cpu-baseline 2> $out/bench.err | tail -1 > $out/bench_avatar_$r.json
python - $out/bench_avatar_$r.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("avatar C3: %.1f frames/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
print({k: round(v, 1) for k, v in d["roofline"]["stage_us_per_step"].items()})
PY
done
