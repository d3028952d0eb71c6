# round 6, GPU box, fifth call: the forward blend's block masks as whole-word plain stores (carry between chunks) against OR atomics
# for every word (rounds 4-5) and no emission at all (timing only); parity; the row hand-over microbenchmark
out=gpurun_out/r6_fifth; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1200 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py tests/test_headline_gpu.py -x -q -m gpu > $out/tests_parity.txt 2>&1
tail -4 $out/tests_parity.txt
bash scripts/ab_variants.sh render_forward fwd_or fwd_nomask 2>&1 | tee $out/ab_forward.txt
timeout 300 scripts/micro/row_handover.bin 2>&1 | tee $out/row_handover.txt
timeout 3000 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1
tail -4 $out/tests.txt
