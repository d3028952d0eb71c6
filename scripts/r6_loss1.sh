# round 6, GPU box: frame_loss without the target normals of unrendered pixels: the loss tests, then the bench
out=gpurun_out/r6_loss1; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_plugin_gpu.py -m gpu -x -q -k "loss or step_plan" 2>&1 | tail -4 | tee $out/tests.txt
for r in 1 2; do python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_us']['frame_loss'])"; done | tee $out/bench.txt
