# round 6, GPU box, fourth call: the forward blend with chunks aligned to the mask words (plain stores) against round 5's placement,
# the emission behind the gather alone, and no emission at all (timing only); parity suites on the new kernel
out=gpurun_out/r6_fourth; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1200 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py tests/test_headline_gpu.py -x -q -m gpu > $out/tests_parity.txt 2>&1
tail -4 $out/tests_parity.txt
bash scripts/ab_variants.sh render_forward fwd_r5 fwd_behind fwd_nomask 2>&1 | tee $out/ab_forward.txt
timeout 3000 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1
tail -4 $out/tests.txt
