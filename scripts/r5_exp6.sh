out=gpurun_out/r5_exp6; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_training_gpu.py tests/test_plugin_gpu.py -x -q -m gpu > $out/tests.txt 2>&1
tail -5 $out/tests.txt
for r in 1 2; do
python bench.py --loss avatar --steps 100 --warmup 5 --no-cpu-baseline 2> $out/avatar.err | tail -1 > $out/avatar.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_exp6/avatar.json"))
print("avatar line: %.1f frames/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
print({k: round(v, 1) for k, v in d["roofline"]["stage_us_per_step"].items()})
PY
done
