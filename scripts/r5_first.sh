# round 5, GPU box: the headline pin test + the driver's exact bench command in five fresh processes
out=gpurun_out/r5_first; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_headline_gpu.py -x -q -m gpu -s > $out/headline.txt 2>&1
tail -15 $out/headline.txt
for r in 1 2 3 4 5; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $out/spread_$r.err | tail -1 > $out/spread_$r.json
  python - $out/spread_$r.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("driver form: %.1f frames/s  %.4f ms/step  host issue %.3f  frac %.4f" % (d["value"], d["ms_per_step"], d["config"]["host_issue_ms_per_step"], d["roofline"]["frac"]))
PY
done
