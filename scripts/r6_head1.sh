# round 6, GPU box: is the fused head (warp + preprocess, 31 us for ~5 us of resident wavefront time) held back by workgroup residency / dispatch?
out=gpurun_out/r6_head1; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
bash scripts/ab_variants.sh lbs_warp_forward head_lds4 head_t2 head_t4 2>&1 | tee $out/ab.txt
