# round 6, GPU box, ninth call: pairs of blocks with the rows of wide splats in float64 (development: -DSOAR_BWD_MIXED=1) --
# the strict C3 surfel bar 8 x per variant, then the stage times
out=gpurun_out/r6_ninth; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for v in mixed2 mixed1 region2; do
  for i in 1 2 3 4 5 6 7 8; do
    SOAR_HIP_LIB=$PWD/soar_amd/_lib/variants/$v.so timeout 600 python -m pytest tests/test_reference_build_gpu.py -q -m gpu -s -k "C3_100k_1080p" 2>&1 | grep -E "^C3_100k|passed|failed" | cut -c1-330 | sed "s/^/$v  /" >> $out/strict_c3.txt
  done
done
cat $out/strict_c3.txt | grep -o "^[a-z0-9]*  C3.*dL_drotations': '[0-9.e-]*'" | sed "s/C3_100k.*dL_drotations/ dL_drotations/" | sort | uniq -c
grep -c passed $out/strict_c3.txt
bash scripts/ab_variants.sh render_backward mixed2 mixed1 region2 2>&1 | tee $out/ab_backward.txt
