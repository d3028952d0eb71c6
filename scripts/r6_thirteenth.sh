# round 6, GPU box, thirteenth call: the fused head of the forward pass (soar_frames_warp_preprocess) -- bit-equality, the suites, timing
out=gpurun_out/r6_thirteenth; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1200 python -m pytest tests/test_plugin_gpu.py -x -q -m gpu -k "fused_head or fused_tail or step_plan" > $out/tests_head.txt 2>&1
tail -5 $out/tests_head.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1
tail -4 $out/tests.txt
for v in 1 0 1 0; do
  SOAR_PLAN_FUSED_HEAD=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us_per_step']
print('fused_head=$v  %.1f frames/s  %.4f ms/step   preprocess %s  lbs_warp_forward %s  depth_order %s' % (d['value'], d['ms_per_step'], s.get('preprocess'), s.get('lbs_warp_forward'), s.get('depth_order')))" | tee -a $out/ab_head.txt
done
