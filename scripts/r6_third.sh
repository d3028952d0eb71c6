# round 6, GPU box, third call: A/B of the backward blend's wait placement (default = new), the old order without the explicit wait,
# round 5's kernel, pairs of blocks; the forward without its mask atomics (timing only); then the full GPU suite on the new kernel
out=gpurun_out/r6_third; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
bash scripts/ab_variants.sh render_backward nowait r5bwd region2 2>&1 | tee $out/ab_backward.txt
bash scripts/ab_variants.sh render_forward fwd_nomask 2>&1 | tee $out/ab_forward.txt
timeout 3000 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1
tail -5 $out/tests.txt
