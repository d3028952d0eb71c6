# round 6, GPU box: blend_frame_matrix reading the weights of 8 joints per LDS round trip: the LBS / plan tests, then A/B (1 = the form before)
out=gpurun_out/r6_lbs2; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests -m gpu -x -q -k "lbs or warp or fused_head or fused_tail or headline" 2>&1 | tail -3 | tee $out/tests.txt
run() { python "$@" --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us']; print('head %5.1f tail %5.1f us  %.3f ms/step' % (s['lbs_warp_forward'], s['lbs_warp_backward'], d['ms_per_step']))"; }
{ echo -n "standard (8) "; run bench.py; for n in lbs_w1 lbs_w4 lbs_w16; do echo -n "$n  "; run scripts/ab_lib.py soar_amd/_lib/variants/$n.so; done; echo -n "standard (8) "; run bench.py; } | tee $out/ab.txt
