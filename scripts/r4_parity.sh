# round 4, GPU box: the parity experiments of VERDICT r3 item 1.  Writes gpurun_out/r4_parity/*
set -x
out=gpurun_out/r4_parity; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python -m pytest tests/test_lbs_gpu.py tests/test_optim_gpu.py tests/test_reference_build_gpu.py -x -q -s -m gpu -k "knn or adam or rows_of or surfel or c5" > $out/tests.txt 2>&1
tail -5 $out/tests.txt
# cost of the division forms on the step
bash scripts/ab_variants.sh render_backward bwd_div1 bwd_div2 > $out/ab_div.txt 2>&1
cat $out/ab_div.txt
# the scenes VERDICT names, three forms of the transmittance reconstruction
for v in standard bwd_div1 bwd_div2; do
  if [ $v = standard ]; then unset SOAR_HIP_LIB; else export SOAR_HIP_LIB=$PWD/soar_amd/_lib/variants/$v.so; fi
  python scripts/gradient_gap.py 209 772 858 934 1316 1388 1498 > $out/gap_$v.txt 2>&1
done
unset SOAR_HIP_LIB
# ratio statistics over all 1500 scenes, two runs at once (the float atomics differ from run to run)
SOAR_FUZZ_THREADS=8 python tests/tools/fuzz_vs_reference.py 1500 70000 --ratios > $out/fuzz_ratios_run1.txt 2>&1 &
p1=$!
SOAR_FUZZ_THREADS=8 python tests/tools/fuzz_vs_reference.py 1500 70000 --ratios > $out/fuzz_ratios_run2.txt 2>&1 &
p2=$!
wait $p1 $p2
tail -40 $out/fuzz_ratios_run1.txt
