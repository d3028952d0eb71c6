# round 6, GPU box: the fused head with the Gaussian's own inputs read in front of the joint-transform blend: bit-equality tests, then timing
out=gpurun_out/r6_head2; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests -m gpu -x -q -k "lbs or warp or fused_head or fused_tail or headline or preprocess or parity" 2>&1 | tail -3 | tee $out/tests.txt
for r in 1 2 3; do python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us']; print('head %5.1f tail %5.1f us  %.3f ms/step' % (s['lbs_warp_forward'], s['lbs_warp_backward'], d['ms_per_step']))"; done | tee $out/bench.txt
