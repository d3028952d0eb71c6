# round 6, GPU box: what the driver runs at round end, on the final tree
out=gpurun_out/r6_final_check; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $out/tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $out/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_driver_form.json; python -c "
import json; d=json.load(open('$out/bench_driver_form.json')); r=d['roofline']; print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['repeats_ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'], d['reference_same_box'])"
