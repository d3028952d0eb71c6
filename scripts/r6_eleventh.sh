# round 6, GPU box, eleventh call: occupancy of the fused tail kernel (3 / 4 / 5 waves per SIMD), the float64-row plan test again
out=gpurun_out/r6_eleventh; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1200 python -m pytest tests/test_plugin_gpu.py -x -q -m gpu -k "fused_tail or step_plan" > $out/tests_tail.txt 2>&1
tail -3 $out/tests_tail.txt
bash scripts/ab_variants.sh lbs_warp_backward tail3 tail5 2>&1 | tee $out/ab_tail_wpe.txt
