# round 5, GPU box: what the driver runs at round end (GPU suite, smoke, the bench line), then the profile set again on the final build
out=gpurun_out/r5_final; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 3000 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1
tail -3 $out/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2> $out/bench.err | tail -1 > $out/bench_driver_form.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_final/bench_driver_form.json"))
r = d["roofline"]
print("driver form:", d["value"], d["ms_per_step"], "frac", r["frac"], "counter_frac", r.get("counter_frac"), "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
bash scripts/r5_profiles.sh r05 > $out/profiles.log 2>&1
tail -30 $out/profiles.log
