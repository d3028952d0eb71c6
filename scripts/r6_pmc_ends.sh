# round 6, GPU box: SQ counters of the fused per-Gaussian ends of a step (what are their 33 + 38 us made of?)
out=gpurun_out/r6_pmc_ends; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
bash scripts/pmc_kernel.sh "warp_preprocess_frames|geom_warp_backward_frames" > $out/sq.txt 2>&1
cat $out/sq.txt
