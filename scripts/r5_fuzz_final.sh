# round 5, GPU box: the 1500-scene fuzz against the reference's own kernels (the seeds of round 4's sweep) on the FINAL kernels of the
# round (blends without the register prefetch, whole-row hand-over, packed pixel steps), ratio statistics included, and the strict
# surfel bars with their measured distances.  Writes gpurun_out/r5_fuzz_final/*
out=gpurun_out/r5_fuzz_final; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python -m pytest tests/test_reference_build_gpu.py -x -q -s -m gpu -k "surfel or c5" > $out/strict_bars.txt 2>&1
tail -12 $out/strict_bars.txt
SOAR_FUZZ_THREADS=8 timeout 2400 python tests/tools/fuzz_vs_reference.py 1500 70000 --ratios > $out/fuzz_ratios.txt 2>&1
tail -45 $out/fuzz_ratios.txt
