# round 6, GPU box: the full GPU suite + smoke + the driver's bench command on the tree with the fused head and tail
out=gpurun_out/r6_fourteenth; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 3000 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1
tail -4 $out/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2> $out/bench.err | tail -1 > $out/bench.json
python - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("driver form: %.1f frames/s  %.4f ms/step repeats %s frac %.4f" % (d["value"], d["ms_per_step"], d["repeats_ms_per_step"], d["roofline"]["frac"]))
print(d["roofline"]["stage_us_per_step"])
PY
