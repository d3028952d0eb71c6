# round 5, GPU box: forward blend with the quad-level cull in front of the blocks' tests -- parity, then time (also at 5 waves / SIMD)
out=gpurun_out/r5_exp2; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py tests/test_headline_gpu.py -x -q -m gpu > $out/tests.txt 2>&1
tail -5 $out/tests.txt
bash scripts/ab_variants.sh render_forward fwd_wpe5
