# round 6, GPU box, tenth call: the fused tail of the backward pass (soar_frames_geometry_warp_backward) -- bit-equality with the two
# kernels it replaces, the suites that run the plan, and the step with and without it
out=gpurun_out/r6_tenth; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1200 python -m pytest tests/test_plugin_gpu.py -x -q -m gpu -k "fused_tail or step_plan" > $out/tests_tail.txt 2>&1
tail -5 $out/tests_tail.txt
timeout 1200 python -m pytest tests/test_headline_gpu.py tests/test_training_gpu.py tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py -x -q -m gpu > $out/tests_more.txt 2>&1
tail -3 $out/tests_more.txt
for v in 1 0 1 0; do
  SOAR_PLAN_FUSED_TAIL=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us_per_step']
print('fused_tail=$v  %.1f frames/s  %.4f ms/step   geometry_backward %s  lbs_warp_backward %s' % (d['value'], d['ms_per_step'], s.get('geometry_backward'), s.get('lbs_warp_backward')))" | tee -a $out/ab_tail.txt
done
