# round 6, GPU box: the 1500-scene fuzz against the reference's own kernels (the seeds of rounds 4-5) on the final kernels of the round
# (the per-Gaussian stages are header functions now: preprocess_point.h / geom_bwd_point.h), ratio statistics included
out=gpurun_out/r6_fuzz_final; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
SOAR_FUZZ_THREADS=8 timeout 2400 python tests/tools/fuzz_vs_reference.py 1500 70000 --ratios > $out/fuzz_ratios.txt 2>&1
tail -30 $out/fuzz_ratios.txt | cut -c1-260
