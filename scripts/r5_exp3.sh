# round 5, GPU box: plugin suite with the early status check, bench line with the counter rates
out=gpurun_out/r5_exp3; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_plugin_gpu.py tests/test_integration_stub_gpu.py tests/test_bench_gpu.py -x -q -m gpu > $out/tests.txt 2>&1
tail -15 $out/tests.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $out/bench.err | tail -1 > $out/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_exp3/bench.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], d["config"]["device_events_ms_per_step"], d["config"]["library"])
print({k: r[k] for k in ("achieved", "frac", "traffic", "counter_GBs", "counter_frac")}, r["whole_frame"])
PY
python scripts/plugin_time.py 2>&1 | tail -12
