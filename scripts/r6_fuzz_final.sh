# round 6, GPU box: the 1500-scene fuzz against the reference's own kernels (the seeds of rounds 4-5) on the final kernels of the round
# (the per-Gaussian stages are header functions now: preprocess_point.h / geom_bwd_point.h; bin_tiles rewritten), ratio statistics included;
# then the first 600 scenes once more with bin_tiles' helpers switched on for every band of more than 16 rectangles
out=gpurun_out/r6_fuzz_final; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
SOAR_FUZZ_THREADS=8 timeout 2400 python tests/tools/fuzz_vs_reference.py 1500 70000 --ratios > $out/fuzz_ratios.txt 2>&1
tail -3 $out/fuzz_ratios.txt | cut -c1-260
SOAR_BIN_SPLIT_AT=16 SOAR_FUZZ_THREADS=8 timeout 2400 python tests/tools/fuzz_vs_reference.py 600 70000 > $out/fuzz_split16.txt 2>&1
tail -2 $out/fuzz_split16.txt | cut -c1-260
