# round 6, GPU box, seventh call: round 5 emission moved behind the gather wait and barrier, tail emission behind the epilogue
out=gpurun_out/r6_seventh; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
bash scripts/ab_variants.sh render_forward fwd_behind fwd_nomask 2>&1 | tee $out/ab_forward.txt
